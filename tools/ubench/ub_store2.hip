// Micro-benchmark: cache-policy flavours of the sampler's 16-byte store (gfx950: sc0 / sc1 / nt bits), as a plain linear fill and in
// the sampling kernel's pattern (workgroup = 8 x rows x 256 z of one y, a wavefront two rows, 1 KiB per wavefront instruction),
// 512^3 floats, four buffers in rotation like bench.py's roofline pass.  Experiment harness, not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int N = 512;
typedef float vf4 __attribute__((ext_vector_type(4)));

template <int F> __device__ __forceinline__ void st(float* p, vf4 v)
{
    if (F == 0) *reinterpret_cast<vf4*>(p) = v;
    else if (F == 1) __builtin_nontemporal_store(v, reinterpret_cast<vf4*>(p));
    else if (F == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else if (F == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    else if (F == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
    else if (F == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
    else if (F == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(p), "v"(v) : "memory");
}
template <int F> __global__ __launch_bounds__(256) void k_linear(float* v)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    st<F>(v + i, vf4{1.f, 2.f, 3.f, (float)i});
}
template <int F> __global__ __launch_bounds__(256) void k_x8(float* v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int z = blockIdx.x * 256 + 4 * lane, iy = blockIdx.y, x8 = blockIdx.z;
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const int ix = x8 * 8 + wave * 2 + rr;
        st<F>(v + ((size_t)ix * N + iy) * N + z, vf4{(float)ix, (float)iy, (float)z, 1.f});
    }
}
template <class L> float timeit(L f, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 8; i++) f(i);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < iters; i++) f(i);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / iters;
}
int main()
{
    float* buf[4];
    const size_t n = (size_t)N * N * N;
    for (auto& b : buf) CK(hipMalloc(&b, n * 4));
    const char* names[8] = {"plain", "nt", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt", "sc0", "sc0 nt"};
    for (int rep = 0; rep < 2; rep++) {
#define RUN(F) { \
        const float a = timeit([&](int i) { hipLaunchKernelGGL(k_linear<F>, dim3(n / 1024), dim3(256), 0, 0, buf[i & 3]); }, 40); \
        const float b = timeit([&](int i) { hipLaunchKernelGGL(k_x8<F>, dim3(N / 256, N, N / 8), dim3(256), 0, 0, buf[i & 3]); }, 40); \
        printf("%-12s linear %6.1f us (%.2f TB/s)   x8 pattern %6.1f us (%.2f TB/s)\n", names[F], a, n * 4 / a * 1e-6, b, n * 4 / b * 1e-6); }
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
    }
    return 0;
}
