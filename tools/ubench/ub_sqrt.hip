// Exhaustive check of candidate correctly-rounded float square roots for the sampling kernels' sdfk_sqrt
// (sample_codegen.h), over EVERY positive normal float: how far is the hardware estimate off, and which short
// correction sequences give the correctly rounded result everywhere.  Experiment harness, not part of the product.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off ub_sqrt.hip -o ub_sqrt
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ float exact_sqrt(float x) { return (float)__builtin_sqrt((double)x); }   // fp64 sqrt is correctly rounded; 53 >= 2*24+2: no double rounding

// 0: v_sqrt_f32 alone          1: x * v_rsq_f32(x) alone
// 2: rsq, one Markstein step   3: rsq, Goldschmidt step + Markstein step (LLVM's flush-mode lowering)
// 4: v_sqrt + upward check only   5: v_sqrt + downward check only   6: the current sdfk_sqrt short path
// 7: v_sqrt, h = 0.5 * v_rcp(s), one Markstein step
template <int V>
__device__ __forceinline__ float cand(float x)
{
    if (V == 0) return __builtin_amdgcn_sqrtf(x);
    if (V == 1) return x * __builtin_amdgcn_rsqf(x);
    if (V == 2) {
        const float y = __builtin_amdgcn_rsqf(x);
        const float g = x * y, h = 0.5f * y;
        const float d = __builtin_fmaf(-g, g, x);
        return __builtin_fmaf(d, h, g);
    }
    if (V == 3) {
        const float y = __builtin_amdgcn_rsqf(x);
        float g = x * y, h = 0.5f * y;
        const float e = __builtin_fmaf(-h, g, 0.5f);
        h = __builtin_fmaf(h, e, h);
        g = __builtin_fmaf(g, e, g);
        const float d = __builtin_fmaf(-g, g, x);
        return __builtin_fmaf(d, h, g);
    }
    float s = __builtin_amdgcn_sqrtf(x);
    if (V == 7) {
        const float h = 0.5f * __builtin_amdgcn_rcpf(s);
        const float d = __builtin_fmaf(-s, s, x);
        return __builtin_fmaf(d, h, s);
    }
    const float sm = __builtin_bit_cast(float, __builtin_bit_cast(int, s) - 1);
    const float sp = __builtin_bit_cast(float, __builtin_bit_cast(int, s) + 1);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    if (V != 4 && rm <= 0.0f) s = sm;
    if (V != 5 && rp > 0.0f) s = sp;
    return s;
}

// hist[0..4]: candidate - exact in ulps: <= -2, -1, 0, +1, >= +2
template <int V>
__global__ void k(unsigned long long* hist, uint32_t lo, uint32_t hi, uint32_t* first_bad)
{
    unsigned long long h[5] = {0, 0, 0, 0, 0};
    for (uint64_t b = (uint64_t)lo + blockIdx.x * 256u + threadIdx.x; b < hi; b += (uint64_t)gridDim.x * 256u) {
        const float x = __builtin_bit_cast(float, (uint32_t)b);
        const int d = __builtin_bit_cast(int, cand<V>(x)) - __builtin_bit_cast(int, exact_sqrt(x));
        const int c = d < -1 ? 0 : d > 1 ? 4 : d + 2;
        h[c]++;
        if (d != 0) atomicMin(first_bad, (uint32_t)b);
    }
    for (int c = 0; c < 5; c++) if (h[c]) atomicAdd(&hist[c], h[c]);
}

template <int V>
int run(const char* name, unsigned long long* d_hist, uint32_t* d_bad, uint32_t lo, uint32_t hi)
{
    CK(hipMemset(d_hist, 0, 5 * sizeof(unsigned long long)));
    CK(hipMemset(d_bad, 0xff, 4));
    hipLaunchKernelGGL(k<V>, dim3(256 * 32), dim3(256), 0, 0, d_hist, lo, hi, d_bad);
    CK(hipDeviceSynchronize());
    unsigned long long h[5]; uint32_t bad;
    CK(hipMemcpy(h, d_hist, sizeof h, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    printf("%-58s  <=-2 %llu  -1 %llu  exact %llu  +1 %llu  >=+2 %llu   first bad 0x%08x\n", name, h[0], h[1], h[2], h[3], h[4], bad);
    return 0;
}

int main()
{
    unsigned long long* d_hist; uint32_t* d_bad;
    CK(hipMalloc(&d_hist, 5 * sizeof(unsigned long long))); CK(hipMalloc(&d_bad, 4));
    const uint32_t ranges[3][2] = {{0x00800000u, 0x7f800000u}, {0x0f800000u, 0x7f800000u}, {0x00000001u, 0x00800000u}};
    const char* rn[3] = {"all positive normals", "x >= 2^-96", "denormals"};
    for (int r = 0; r < 3; r++) {
        printf("== %s\n", rn[r]);
        const uint32_t lo = ranges[r][0], hi = ranges[r][1];
        if (run<0>("v_sqrt_f32", d_hist, d_bad, lo, hi)) return 1;
        if (run<1>("x * v_rsq_f32(x)", d_hist, d_bad, lo, hi)) return 1;
        if (run<2>("rsq + one Markstein step (5 ops)", d_hist, d_bad, lo, hi)) return 1;
        if (run<3>("rsq + Goldschmidt + Markstein (8 ops)", d_hist, d_bad, lo, hi)) return 1;
        if (run<4>("v_sqrt + upward check only", d_hist, d_bad, lo, hi)) return 1;
        if (run<5>("v_sqrt + downward check only", d_hist, d_bad, lo, hi)) return 1;
        if (run<6>("v_sqrt + both checks (current short path, 9 ops)", d_hist, d_bad, lo, hi)) return 1;
        if (run<7>("v_sqrt + rcp + one Markstein step (5 ops, 2 transc.)", d_hist, d_bad, lo, hi)) return 1;
    }
    return 0;
}
