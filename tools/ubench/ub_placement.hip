// Micro-benchmark (round 6): is the colour sampler's rate a property of WHERE its two arrays lie?  The same binary samples the README scene
// in 320 us on one box and 350-385 us on others -- and the store pattern alone (ub_mixstore.hip) shows 0.72 and 0.75 on two boxes.  Here:
// the sampler's store pattern (8 x rows x 256 z per workgroup, value KiB + three colour KiB per row) into EIGHT different pairs of
// allocations of one process, each timed alone -- then the same with both arrays carved out of one allocation at several relative offsets.
// Experiment harness, not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int N = 512;
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st(float* p, vf4 v) { __builtin_nontemporal_store(v, reinterpret_cast<vf4*>(p)); }
__global__ __launch_bounds__(256) void k_tile_vc(float* v, float* c)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int rr = 0; rr < 2; rr++) {
        const int ix = blockIdx.z * 8 + wave * 2 + rr;
        const size_t o = ((size_t)ix * N + blockIdx.y) * N + blockIdx.x * 256;
        st(v + o + 4 * lane, vf4{1.f, 2.f, 3.f, 4.f});
        for (int q = 0; q < 3; q++) st(c + o * 3 + 256 * q + 4 * lane, vf4{(float)q, 2.f, 3.f, 4.f});
    }
}
__global__ __launch_bounds__(256) void k_fill(float* c)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    st(c + i, vf4{1.f, 2.f, 3.f, 4.f});
}
static float time_pair(float* v, float* c, int iters = 20)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 30; i++) hipLaunchKernelGGL(k_tile_vc, dim3(N / 256, N, N / 8), dim3(256), 0, 0, v, c);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k_tile_vc, dim3(N / 256, N, N / 8), dim3(256), 0, 0, v, c);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms * 1000.f / iters;
}
int main()
{
    const size_t nvox = (size_t)N * N * N, B = nvox * 16;
    std::vector<float*> vs, cs;
    for (int t = 0; t < 8; t++) {
        float *v, *c;
        CK(hipMalloc(&v, nvox * 4)); CK(hipMalloc(&c, nvox * 12));
        vs.push_back(v); cs.push_back(c);
        const float us = time_pair(v, c);
        printf("pair %d: values %p colours %p  (c - v = %+.1f MiB)   %6.1f us  %.3f of 8 TB/s\n", t, (void*)v, (void*)c, ((char*)c - (char*)v) / 1048576.0, us, B / us * 1e-6 / 8);
    }
    // cross pairs: values of pair i with colours of pair j
    for (int t = 0; t < 8; t++) {
        const int i = t, j = (t + 3) % 8;
        const float us = time_pair(vs[i], cs[j]);
        printf("values %d + colours %d: %6.1f us  %.3f\n", i, j, us, B / us * 1e-6 / 8);
    }
    for (auto p : vs) hipFree(p);
    for (auto p : cs) hipFree(p);
    // one allocation, the colours at chosen offsets behind the values
    char* big;
    CK(hipMalloc(&big, B + (size_t(64) << 20)));
    for (size_t off : {(size_t)0, (size_t)4096, (size_t)65536, (size_t)(1 << 20), (size_t)(2 << 20) + 4096, (size_t)(3 << 20), (size_t)(17 << 20) + 8192, (size_t)(32 << 20)}) {
        float* v = (float*)big;
        float* c = (float*)(big + nvox * 4 + off);
        const float us = time_pair(v, c);
        printf("one allocation, colours %8zu B behind the end of the values: %6.1f us  %.3f\n", off, us, B / us * 1e-6 / 8);
    }
    // and the same allocation again after a pause / in another order (is it stable in time?)
    for (int rep = 0; rep < 4; rep++) {
        const float us = time_pair((float*)big, (float*)(big + nvox * 4));
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_fill, dim3(nvox * 3 / 1024), dim3(256), 0, 0, (float*)(big + nvox * 4));
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("again: tile %6.1f us (%.3f); plain fill of the colour part %6.1f us (%.3f)\n", us, B / us * 1e-6 / 8, ms * 50.f, nvox * 12 / (ms * 50.f) * 1e-6 / 8);
    }
    return 0;
}
