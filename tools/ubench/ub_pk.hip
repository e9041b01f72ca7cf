// Issue rate and bit-exactness of the packed f32 VALU operations (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) against their
// scalar forms on gfx950: would evaluating two voxels per instruction in the sampling kernels (sample_codegen.h) save vector
// issue slots, and does it give the same bits (denormal operands and results included)?  Experiment harness, not product code.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off ub_pk.hip -o ub_pk
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

// MODE 0: 16 independent v_fma_f32 per iteration   1: 16 v_pk_fma_f32 (32 fmas)   2: 16 v_mul_f32   3: 16 v_pk_mul_f32
//      4: 16 v_add_f32   5: 16 v_pk_add_f32   6: 8 v_rsq_f32   7: mix of the sampler's short sqrt, scalar   8: the same, packed
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters, float seed)
{
    float a[16]; f2 p[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = seed + (float)(threadIdx.x + i); p[i] = f2{a[i], a[i] + 0.5f}; }
    const float m = 1.0000001f, c = 1e-9f;
    const f2 pm = {m, m}, pc = {c, c};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
            if (MODE == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
            if (MODE == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (MODE == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            if (MODE == 6 && i < 8) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
        }
        if (MODE == 7) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                float x = a[i], y, g, h, d;
                asm volatile("v_rsq_f32 %0, %1" : "=v"(y) : "v"(x));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(g) : "v"(x), "v"(y));
                asm volatile("v_mul_f32 %0, 0.5, %1" : "=v"(h) : "v"(y));
                asm volatile("v_fma_f32 %0, -%1, %1, %2" : "=v"(d) : "v"(g), "v"(x));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(d), "v"(h), "v"(g));
            }
        }
        if (MODE == 8) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                f2 x = p[i], y, g, h, d;
                const f2 half = {0.5f, 0.5f};
                asm volatile("v_rsq_f32 %0, %1" : "=v"(y.x) : "v"(x.x));
                asm volatile("v_rsq_f32 %0, %1" : "=v"(y.y) : "v"(x.y));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(g) : "v"(x), "v"(y));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(h) : "v"(half), "v"(y));
                asm volatile("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(g), "v"(x));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p[i]) : "v"(d), "v"(h), "v"(g));
            }
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
}

// bit-exactness: scalar vs packed results of mul / add / fma on operand pairs that include denormals, huge values, NaN, inf
__global__ void k_exact(const float* __restrict__ xs, int n, unsigned* bad, unsigned* examples)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * n) return;
    const float a = xs[i / n], b = xs[i % n], c = xs[(i * 7 + 3) % n];
    float sm, sa, sf, sn;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(sm) : "v"(a), "v"(b));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(sa) : "v"(a), "v"(b));
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(sf) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(sn) : "v"(a), "v"(b), "v"(c));
    const f2 pa = {a, b}, pb = {b, a}, pcc = {c, c};
    f2 pm_, pa_, pf_, pn_;
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pm_) : "v"(pa), "v"(pb));
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(pa_) : "v"(pa), "v"(pb));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(pf_) : "v"(pa), "v"(pb), "v"(pcc));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(pn_) : "v"(pa), "v"(pb), "v"(pcc));
    auto bits = [](float f) { return __builtin_bit_cast(unsigned, f); };
    // (NaN payloads may differ between the forms: compare NaN-ness)
    auto same = [&](float p, float q) { return bits(p) == bits(q) || (p != p && q != q); };
    unsigned w = 0;
    if (!same(sm, pm_.x) || !same(sm, pm_.y)) w |= 1;
    if (!same(sa, pa_.x) || !same(sa, pa_.y)) w |= 2;
    if (!same(sf, pf_.x) || !same(sf, pf_.y)) w |= 4;
    if (!same(sn, pn_.x) || !same(sn, pn_.y)) w |= 8;
    if (w) {
        const unsigned k = atomicAdd(bad, 1u);
        if (k < 8) { examples[4 * k] = bits(a); examples[4 * k + 1] = bits(b); examples[4 * k + 2] = bits(c); examples[4 * k + 3] = w; }
    }
}

// The guard of the pair form of the sampler's short square root (sdfk_sqrt2, sample_codegen.h): "x * 2^-30 is a positive normal"
// must be the same predicate as "2^-96 <= x < inf" on EVERY float bit pattern, and the short sequence in packed arithmetic must
// give the same bits as in scalar arithmetic.  counts[0] = patterns that pass, [1] = the two predicates disagree,
// [2] = packed != scalar, [3] = passing patterns whose short root is not the fp64 root rounded once
__global__ __launch_bounds__(256) void k_sqrt_guard(unsigned long long* counts, unsigned* first_bad)
{
    unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * 256u + threadIdx.x; b < (1ull << 32); b += (uint64_t)gridDim.x * 256u) {
        const float x = __builtin_bit_cast(float, (uint32_t)b);
        const float y = __builtin_amdgcn_rsqf(x);
        const float g = x * y, h = 0.5f * y;
        const float d = __builtin_fmaf(-g, g, x);
        const float r = __builtin_fmaf(d, h, g);
        const float x2 = __builtin_bit_cast(float, (uint32_t)b ^ 0x00012345u);   // the pair's other operand
        const f2 px = {x, x2}, py = {y, __builtin_amdgcn_rsqf(x2)};
        const f2 pt = px * 0x1p-30f;
        const f2 pg = px * py, ph = 0.5f * py;
        const f2 pd = __builtin_elementwise_fma(-pg, pg, px);
        const f2 pr = __builtin_elementwise_fma(pd, ph, pg);
        const bool pass = __builtin_amdgcn_classf(pt.x, 0x100);
        const bool ref = (x >= 0x1p-96f) & (x < __builtin_inff());
        const float want = (float)__builtin_sqrt((double)x);
        if (pass) c0++;
        if (pass != ref) { c1++; atomicMin(first_bad, (uint32_t)b); }
        if (pass && __builtin_bit_cast(uint32_t, pr.x) != __builtin_bit_cast(uint32_t, want)) c3++;
        if (__builtin_bit_cast(uint32_t, pr.x) != __builtin_bit_cast(uint32_t, r) && !(pr.x != pr.x && r != r)) c2++;
    }
    if (c0) atomicAdd(&counts[0], c0);
    if (c1) atomicAdd(&counts[1], c1);
    if (c2) atomicAdd(&counts[2], c2);
    if (c3) atomicAdd(&counts[3], c3);
}

template <int MODE>
int rate(const char* name, float* d_out, double ops_per_iter)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int waves = 1; waves <= 8; waves *= 2) {   // waves per SIMD: blocks of 256 = 1 wave per SIMD of a CU
        const int blocks = 256 * waves;
        hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 100, 1.0f);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0f);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        // wave-instructions per SIMD = waves * iters * ops_per_iter; cycles at 2.4 GHz
        const double cyc = ms * 1e-3 * 2.4e9;
        printf("%-34s %d wave(s)/SIMD: %8.3f ms  -> %.2f cycles @2.4GHz per wave-instruction\n", name, waves, ms, cyc / (waves * (double)iters * ops_per_iter));
    }
    return 0;
}

int main()
{
    float* d_out; CK(hipMalloc(&d_out, 64));
    if (rate<0>("v_fma_f32", d_out, 16)) return 1;
    if (rate<1>("v_pk_fma_f32", d_out, 16)) return 1;
    if (rate<2>("v_mul_f32", d_out, 16)) return 1;
    if (rate<3>("v_pk_mul_f32", d_out, 16)) return 1;
    if (rate<4>("v_add_f32", d_out, 16)) return 1;
    if (rate<5>("v_pk_add_f32", d_out, 16)) return 1;
    if (rate<6>("v_rsq_f32", d_out, 8)) return 1;
    if (rate<7>("short sqrt x16 scalar (5 ops each)", d_out, 16)) return 1;
    if (rate<8>("short sqrt x16 packed (8 x 6 ops)", d_out, 16)) return 1;
    // exactness
    std::vector<float> xs;
    auto fb = [](uint32_t u) { return __builtin_bit_cast(float, u); };
    const uint32_t pats[] = {0x00000000u, 0x00000001u, 0x00000002u, 0x00400000u, 0x007fffffu, 0x00800000u, 0x00800001u, 0x00ffffffu, 0x01000000u,
                             0x0c000000u, 0x1f800000u, 0x20000000u, 0x33800000u, 0x34000000u, 0x3f000000u, 0x3f7fffffu, 0x3f800000u, 0x3f800001u,
                             0x3fffffffu, 0x40490fdbu, 0x4b000000u, 0x4b800001u, 0x5f000000u, 0x5f7fffffu, 0x7e800000u, 0x7f000000u, 0x7f7fffffu,
                             0x7f800000u, 0x7fc00000u, 0x7f800001u};
    for (uint32_t p : pats) { xs.push_back(fb(p)); xs.push_back(fb(p | 0x80000000u)); }
    uint32_t s = 12345u;
    for (int i = 0; i < 1940; i++) { s = s * 1664525u + 1013904223u; xs.push_back(fb(s)); }   // random bit patterns
    const int n = (int)xs.size();
    float* d_x; unsigned *d_bad, *d_ex;
    CK(hipMalloc(&d_x, n * 4)); CK(hipMalloc(&d_bad, 4)); CK(hipMalloc(&d_ex, 128));
    CK(hipMemcpy(d_x, xs.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemset(d_bad, 0, 4)); CK(hipMemset(d_ex, 0, 128));
    hipLaunchKernelGGL(k_exact, dim3((n * n + 255) / 256), dim3(256), 0, 0, d_x, n, d_bad, d_ex);
    CK(hipDeviceSynchronize());
    unsigned bad, ex[32];
    CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ex, d_ex, 128, hipMemcpyDeviceToHost));
    printf("exactness: %d x %d operand pairs (denormals, extremes, NaN/inf, random patterns): %u differ between scalar and packed\n", n, n, bad);
    for (unsigned k = 0; k < (bad < 8 ? bad : 8); k++) printf("  a=0x%08x b=0x%08x c=0x%08x ops=%x\n", ex[4 * k], ex[4 * k + 1], ex[4 * k + 2], ex[4 * k + 3]);
    {
        unsigned long long* d_c; unsigned* d_fb;
        CK(hipMalloc(&d_c, 32)); CK(hipMalloc(&d_fb, 4)); CK(hipMemset(d_c, 0, 32)); CK(hipMemset(d_fb, 0xff, 4));
        hipLaunchKernelGGL(k_sqrt_guard, dim3(256 * 32), dim3(256), 0, 0, d_c, d_fb);
        CK(hipDeviceSynchronize());
        unsigned long long c[4]; unsigned fbad;
        CK(hipMemcpy(c, d_c, 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(&fbad, d_fb, 4, hipMemcpyDeviceToHost));
        printf("pair sqrt guard over all 2^32 patterns: %llu pass, %llu disagree with 2^-96 <= x < inf (first 0x%08x), %llu packed != scalar, %llu passing roots wrong\n",
               c[0], c[1], fbad, c[2], c[3]);
    }
    return 0;
}
