// Micro-benchmark of the ordered compaction (k_compact) on a 512^3 sphere; experiment harness,
// not part of the product.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off ub_compact.hip -o ub_compact
#include "../../sdfkit_amd/csrc/mc_kernels.hip"
#include <cstdio>
#include <cstring>
#include <vector>
using namespace sdfk;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_sphere(float* v, int n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t tot = (size_t)n * n * n;
    if (i >= tot) return;
    const int z = i % n, y = (i / n) % n, x = i / ((size_t)n * n);
    const float s = 3.0f / (n - 1);
    const float px = -1.5f + x * s, py = -1.5f + y * s, pz = -1.5f + z * s;
    v[i] = sqrtf(px * px + py * py + pz * pz) - 1.0f;
}

__global__ __launch_bounds__(256) void k_empty(McParams P) { if (P.nx == -1) P.counters->pad[0] = 1; }

__global__ __launch_bounds__(256) void k_loadonly(McParams P)
{
    const int b = blockIdx.x;
    const int lay = b / P.bpl;
    const int z = P.lay_count_begin + lay;
    const int nseg = P.ncy * P.nxw;
    const int i0 = (b - lay * P.bpl) * 1024 + 4 * (int)threadIdx.x;
    uint64_t acc = 0;
    if (i0 < nseg) {
        const size_t plane = (size_t)P.ny * P.nxw;
        const uint64_t* f0 = P.bits + (size_t)z * plane + i0;
        uint64_t wa[5], wb[5], wc[5], wd[5];
        load_words5(f0, wa); load_words5(f0 + P.nxw, wb); load_words5(f0 + plane, wc); load_words5(f0 + plane + P.nxw, wd);
        for (int k = 0; k < 5; k++) acc += wa[k] ^ wb[k] ^ wc[k] ^ wd[k];
    }
    if (acc == 0x1234567ull) P.blockcnt[b] = acc;
}

template <class F>
float time_it(F f, int iters = 50)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; i++) f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < iters; i++) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.0f / iters;
}

int main()
{
    const int n = 512;
    const size_t nv = (size_t)n * n * n;
    float* values; CK(hipMalloc(&values, nv * 4));
    hipLaunchKernelGGL(k_sphere, dim3((nv + 255) / 256), dim3(256), 0, 0, values, n);
    McParams P; memset(&P, 0, sizeof P);
    P.values = values; P.nx = P.ny = P.nz = n; P.ncx = P.ncy = P.ncz = n - 1; P.nxw = n / 64;
    P.lay_count_begin = 0; P.lay_emit_begin = 0; P.lay_emit_end = n - 1; P.lay_list_end = n - 1;
    P.iso = 0; P.step = 1;
    uint64_t* bits; CK(hipMalloc(&bits, ((size_t)n * n * P.nxw + 8) * 8));
    P.bits = bits;
    P.bpl = (P.ncy * P.nxw + 1023) / 1024;
    const int nlog = (n - 1) * P.bpl;
    CK(hipMalloc(&P.blockcnt, (nlog + 1) * 8));
    CK(hipMalloc(&P.wavecnt, (nlog + 1) * 16));
    CK(hipMalloc(&P.rowstart, ((size_t)(n - 1) * P.ncy + 2) * 4));
    P.cap_active = 1u << 20;
    CK(hipMalloc(&P.rec_xy, P.cap_active * 4)); CK(hipMalloc(&P.rec_z, P.cap_active * 4));
    CK(hipMalloc(&P.counters, sizeof(McCounters)));
    uint8_t* bits8; CK(hipMalloc(&bits8, (size_t)n * (n / 8) * n + 64));
    hipLaunchKernelGGL(k_signbits8<false>, dim3((n + 255) / 256, n, n / 8), dim3(256), 0, 0, values, bits8, n, n, n, n / 8, n, 0.0f);
    hipLaunchKernelGGL(k_bits_transpose, dim3((n + 127) / 128, n, (P.nxw + 7) / 8), dim3(256), 0, 0, bits8, bits, n / 8, n, n, P.nxw, n);
    CK(hipDeviceSynchronize());
    printf("blocks %d\n", nlog);
    printf("empty      %7.2f us\n", time_it([&] { hipLaunchKernelGGL(k_empty, dim3(nlog), dim3(256), 0, 0, P); }));
    printf("loadonly   %7.2f us\n", time_it([&] { hipLaunchKernelGGL(k_loadonly, dim3(nlog), dim3(256), 0, 0, P); }));
    printf("count      %7.2f us\n", time_it([&] { hipLaunchKernelGGL(k_compact<false>, dim3(nlog), dim3(256), 0, 0, P); }));
    printf("write      %7.2f us\n", time_it([&] { hipLaunchKernelGGL(k_compact<true>, dim3(nlog), dim3(256), 0, 0, P); }));
    printf("count+write%7.2f us\n", time_it([&] { hipLaunchKernelGGL(k_compact<false>, dim3(nlog), dim3(256), 0, 0, P);
                                                   hipLaunchKernelGGL(k_compact<true>, dim3(nlog), dim3(256), 0, 0, P); }));
    McCounters c; CK(hipMemcpy(&c, P.counters, sizeof c, hipMemcpyDeviceToHost));
    printf("n_active %u n13 %u\n", c.n_active, c.n_case13);
    return 0;
}
