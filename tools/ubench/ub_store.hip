// Micro-benchmark of volume store patterns for the fused sampling kernel (512^3 floats, layout
// [x][y][z], z fastest).  Experiment harness, not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int N = 512;

__device__ __forceinline__ float val(int x, int y, int z) { return (float)z * 0.01f - 1.0f + (float)(x + y) * 1e-6f; }

template <bool NT>
__device__ __forceinline__ void st4(float* p, float4 v)
{
    typedef float vf4 __attribute__((ext_vector_type(4)));
    if (NT) { vf4 t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<vf4*>(p)); }
    else *reinterpret_cast<float4*>(p) = v;
}

// V0: linear
template <bool NT>
__global__ __launch_bounds__(256) void k_linear(float* v)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    st4<NT>(v + i, make_float4(1.f, 2.f, 3.f, (float)i));
}

// linear with the chunk index shifted against the workgroup index (XCD <-> channel affinity?)
template <int CHUNK_F4 /* float4 per lane */>
__global__ __launch_bounds__(256) void k_linear_shift(float* v, unsigned shift, unsigned nblk)
{
    const size_t c = (blockIdx.x + shift) % nblk;
    float* p = v + c * (1024 * CHUNK_F4);
    for (int k = 0; k < CHUNK_F4; k++)
        st4<false>(p + (k * 256 + threadIdx.x) * 4, make_float4(1.f, 2.f, 3.f, (float)k));
}

// 1 store per lane, chunk order scrambled: chunk = (blockIdx * mul) % nblk  (mul odd)
__global__ __launch_bounds__(256) void k_linear_scramble(float* v, unsigned mul, unsigned nblk)
{
    const size_t c = ((size_t)blockIdx.x * mul) % nblk;
    st4<false>(v + c * 1024 + threadIdx.x * 4, make_float4(1.f, 2.f, 3.f, 4.f));
}
// THREADS lanes per workgroup, 1 store per lane, contiguous
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_linear_t(float* v)
{
    st4<false>(v + ((size_t)blockIdx.x * THREADS + threadIdx.x) * 4, make_float4(1.f, 2.f, 3.f, 4.f));
}
// 2 stores per lane, the second `far` floats away
__global__ __launch_bounds__(256) void k_linear_2far(float* v, size_t far)
{
    float* p = v + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    st4<false>(p, make_float4(1.f, 2.f, 3.f, 4.f));
    st4<false>(p + far, make_float4(1.f, 2.f, 3.f, 4.f));
}
// 1 store per lane; wave w of the workgroup writes 1 KiB at x-plane stride (like the tile, 1 store)
__global__ __launch_bounds__(256) void k_xstride1(float* v)
{
    // workgroup = (zc half, y, x-group of 4): wave w -> x = 4*xg + w
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int zc = blockIdx.x * 256, iy = blockIdx.y, ix = blockIdx.z * 4 + wave;
    st4<false>(v + ((size_t)ix * N + iy) * N + zc + 4 * lane, make_float4(1.f, 2.f, 3.f, 4.f));
}

// x8 tile: 8 x-rows x 256 z of one y per workgroup; 512 lanes (1 store each) or 256 lanes (2 stores);
// sign bits leave as BYTES (8 x-bits) in bitsT[y][x8][z], z contiguous: 256 B coalesced per workgroup
template <int THREADS, bool NT>
__global__ __launch_bounds__(THREADS) void k_x8(float* v, uint8_t* bitsT)
{
    __shared__ unsigned char nib[8][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int zc = blockIdx.x * 256, iy = blockIdx.y, x8 = blockIdx.z;
    const int z = zc + 4 * lane;
    for (int r = wave; r < 8; r += THREADS / 64) {
        const int ix = x8 * 8 + r;
        float w[4];
        for (int k = 0; k < 4; k++) w[k] = val(ix, iy, z + k);
        st4<NT>(v + ((size_t)ix * N + iy) * N + z, make_float4(w[0], w[1], w[2], w[3]));
        nib[r][lane] = (unsigned char)((w[0] > 0 ? 1u : 0u) | (w[1] > 0 ? 2u : 0u) | (w[2] > 0 ? 4u : 0u) | (w[3] > 0 ? 8u : 0u));
    }
    __syncthreads();
    if (wave == 0) {   // lane = 4 consecutive z: byte k = bit k of the 8 rows' nibbles
        unsigned out = 0;
        for (int r = 0; r < 8; r++) {
            const unsigned n = nib[r][lane];
            out |= ((n & 1u) << r) | (((n >> 1) & 1u) << (8 + r)) | (((n >> 2) & 1u) << (16 + r)) | (((n >> 3) & 1u) << (24 + r));
        }
        *reinterpret_cast<unsigned*>(bitsT + ((size_t)iy * (N / 8) + x8) * N + z) = out;
    }
}

// bitsT[y][x8][z] (bytes) -> bits[z][y][xw] (u64 = the 8 bytes x8 = 8*xw..8*xw+7): workgroup = (y, 128 z)
__global__ __launch_bounds__(256) void k_bits_transpose(const uint8_t* __restrict__ bitsT, uint64_t* __restrict__ bits)
{
    __shared__ uint8_t t[64][128 + 4];
    const int iy = blockIdx.y, z0 = blockIdx.x * 128;
    // 64 rows (x8) x 128 B: 2048 dwords, 8 per lane
    for (int k = threadIdx.x; k < 64 * 32; k += 256) {
        const int row = k >> 5, c = (k & 31) * 4;
        *reinterpret_cast<unsigned*>(&t[row][c]) = *reinterpret_cast<const unsigned*>(bitsT + ((size_t)iy * 64 + row) * N + z0 + c);
    }
    __syncthreads();
    // 128 z x 8 words: 1024 words, 4 per lane; consecutive lanes -> consecutive xw of one z (64 B lines)
    for (int k = threadIdx.x; k < 128 * 8; k += 256) {
        const int zz = k >> 3, xw = k & 7;
        uint64_t w = 0;
        for (int b = 0; b < 8; b++) w |= (uint64_t)t[xw * 8 + b][zz] << (8 * b);
        bits[((size_t)(z0 + zz) * N + iy) * 8 + xw] = w;
    }
}

// V1: current tile: block (zc in 256s, y, xw); wave w rows r=w,w+4..; lane = 4 z
template <bool NT, bool BITS>
__global__ __launch_bounds__(256) void k_cur(float* v, uint64_t* bits)
{
    constexpr int PITCH = 68;
    __shared__ unsigned char nib[64 * PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int zc = blockIdx.x * 256, iy = blockIdx.y, xw = blockIdx.z;
    const int z = zc + 4 * lane;
    for (int r = wave; r < 64; r += 4) {
        const int ix = xw * 64 + r;
        float w[4];
        for (int k = 0; k < 4; k++) w[k] = val(ix, iy, z + k);
        st4<NT>(v + ((size_t)ix * N + iy) * N + z, make_float4(w[0], w[1], w[2], w[3]));
        if (BITS) {
            const unsigned n = (w[0] > 0 ? 1u : 0u) | (w[1] > 0 ? 2u : 0u) | (w[2] > 0 ? 4u : 0u) | (w[3] > 0 ? 8u : 0u);
            nib[r * PITCH + lane] = (unsigned char)n;
        }
    }
    if (BITS) {
        __syncthreads();
        for (int q = wave; q < 64; q += 4) {
            const unsigned n = nib[lane * PITCH + q];
            const unsigned long long b0 = __ballot(n & 1u), b1 = __ballot(n & 2u), b2 = __ballot(n & 4u), b3 = __ballot(n & 8u);
            const int zq = zc + 4 * q;
            if (lane < 4) {
                const unsigned long long wd = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3;
                bits[((size_t)(zq + lane) * N + iy) * (N / 64) + xw] = wd;
            }
        }
    }
}

// generic x-strided tile: 64 x-rows x TZ z of one y; THREADS lanes, each 4 z per pass
template <int TZ, int THREADS, bool NT>
__global__ __launch_bounds__(THREADS) void k_tile(float* v)
{
    constexpr int LPR = TZ / 4;            // lanes per row piece
    constexpr int RPP = THREADS / LPR;     // rows per pass
    const int zt = blockIdx.x * TZ, iy = blockIdx.y, xw = blockIdx.z;
    const int zl = (threadIdx.x % LPR) * 4, r0 = threadIdx.x / LPR;
    for (int r = r0; r < 64; r += RPP) {
        const int ix = xw * 64 + r, z = zt + zl;
        st4<NT>(v + ((size_t)ix * N + iy) * N + z, make_float4(val(ix, iy, z), val(ix, iy, z + 1), val(ix, iy, z + 2), val(ix, iy, z + 3)));
    }
}

// V3: block (ygroup of YT, xw): full z; lane handles z = 4*lane and 4*lane+256
template <bool NT, int YT>
__global__ __launch_bounds__(256) void k_fullz(float* v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int y0 = blockIdx.x * YT, xw = blockIdx.y;
    for (int r = wave; r < 64; r += 4) {
        const int ix = xw * 64 + r;
        for (int yy = 0; yy < YT; yy++) {
            float* row = v + ((size_t)ix * N + y0 + yy) * N;
            for (int h = 0; h < 2; h++) {
                const int z = h * 256 + 4 * lane;
                st4<NT>(row + z, make_float4(val(ix, y0 + yy, z), val(ix, y0 + yy, z + 1), val(ix, y0 + yy, z + 2), val(ix, y0 + yy, z + 3)));
            }
        }
    }
}

// V4: x-major blocks: block = (x, ygroup of 16 rows): 32 KiB contiguous per block
template <bool NT>
__global__ __launch_bounds__(256) void k_rows(float* v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t row0 = (size_t)blockIdx.x * 16;
    for (int r = wave; r < 16; r += 4)
        for (int h = 0; h < 2; h++) {
            const int z = h * 256 + 4 * lane;
            st4<NT>(v + (row0 + r) * N + z, make_float4(1.f, 2.f, 3.f, (float)z));
        }
}

template <class F>
float time_it(F f, int iters = 20)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) f();
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
    const size_t nv = (size_t)N * N * N;
    float* v; CK(hipMalloc(&v, nv * 4));
    uint64_t* bits; CK(hipMalloc(&bits, nv / 8 + 64));
#define RUN(name, ...) printf("%-28s %7.2f us\n", name, time_it([&] { __VA_ARGS__; }))
    RUN("linear", hipLaunchKernelGGL(k_linear<false>, dim3(nv / 1024), dim3(256), 0, 0, v));
    RUN("linear nt", hipLaunchKernelGGL(k_linear<true>, dim3(nv / 1024), dim3(256), 0, 0, v));
    RUN("cur store-only", hipLaunchKernelGGL((k_cur<false, false>), dim3(2, N, N / 64), dim3(256), 0, 0, v, bits));
    RUN("cur store-only nt", hipLaunchKernelGGL((k_cur<true, false>), dim3(2, N, N / 64), dim3(256), 0, 0, v, bits));
    RUN("cur +bits", hipLaunchKernelGGL((k_cur<false, true>), dim3(2, N, N / 64), dim3(256), 0, 0, v, bits));
    RUN("cur +bits nt", hipLaunchKernelGGL((k_cur<true, true>), dim3(2, N, N / 64), dim3(256), 0, 0, v, bits));
    RUN("fullz y1", hipLaunchKernelGGL((k_fullz<false, 1>), dim3(N, N / 64), dim3(256), 0, 0, v));
    RUN("fullz y1 nt", hipLaunchKernelGGL((k_fullz<true, 1>), dim3(N, N / 64), dim3(256), 0, 0, v));
    RUN("fullz y2", hipLaunchKernelGGL((k_fullz<false, 2>), dim3(N / 2, N / 64), dim3(256), 0, 0, v));
    RUN("fullz y4", hipLaunchKernelGGL((k_fullz<false, 4>), dim3(N / 4, N / 64), dim3(256), 0, 0, v));
    RUN("fullz y4 nt", hipLaunchKernelGGL((k_fullz<true, 4>), dim3(N / 4, N / 64), dim3(256), 0, 0, v));
    RUN("rows16", hipLaunchKernelGGL(k_rows<false>, dim3(N * N / 16), dim3(256), 0, 0, v));
    RUN("rows16 nt", hipLaunchKernelGGL(k_rows<true>, dim3(N * N / 16), dim3(256), 0, 0, v));
    uint8_t* bitsT = (uint8_t*)bits + (nv / 8 + 64) / 2;   // (harness only: reuse the allocation)
    CK(hipMalloc(&bitsT, nv / 8 + 64));
    RUN("x8 t512", hipLaunchKernelGGL((k_x8<512, false>), dim3(2, N, N / 8), dim3(512), 0, 0, v, bitsT));
    RUN("x8 t512 nt", hipLaunchKernelGGL((k_x8<512, true>), dim3(2, N, N / 8), dim3(512), 0, 0, v, bitsT));
    RUN("x8 t256", hipLaunchKernelGGL((k_x8<256, false>), dim3(2, N, N / 8), dim3(256), 0, 0, v, bitsT));
    RUN("x8 t256 nt", hipLaunchKernelGGL((k_x8<256, true>), dim3(2, N, N / 8), dim3(256), 0, 0, v, bitsT));
    RUN("bits transpose", hipLaunchKernelGGL(k_bits_transpose, dim3(N / 128, N), dim3(256), 0, 0, bitsT, bits));
    RUN("x8 t512 + transpose", hipLaunchKernelGGL((k_x8<512, false>), dim3(2, N, N / 8), dim3(512), 0, 0, v, bitsT);
        hipLaunchKernelGGL(k_bits_transpose, dim3(N / 128, N), dim3(256), 0, 0, bitsT, bits));
#define TILE(TZ, TH) RUN("tile z" #TZ " t" #TH " nt", hipLaunchKernelGGL((k_tile<TZ, TH, true>), dim3(N / TZ, N, N / 64), dim3(TH), 0, 0, v)); \
                     RUN("tile z" #TZ " t" #TH "   ", hipLaunchKernelGGL((k_tile<TZ, TH, false>), dim3(N / TZ, N, N / 64), dim3(TH), 0, 0, v))
    TILE(256, 256); TILE(256, 512); TILE(256, 1024); TILE(512, 1024); TILE(128, 256); TILE(128, 512); TILE(128, 1024);
    TILE(64, 256); TILE(64, 1024); TILE(512, 256); TILE(512, 512);
    RUN("linear t64", hipLaunchKernelGGL(k_linear_t<64>, dim3(nv / 256), dim3(64), 0, 0, v));
    RUN("linear t128", hipLaunchKernelGGL(k_linear_t<128>, dim3(nv / 512), dim3(128), 0, 0, v));
    RUN("linear t512", hipLaunchKernelGGL(k_linear_t<512>, dim3(nv / 2048), dim3(512), 0, 0, v));
    RUN("linear t1024", hipLaunchKernelGGL(k_linear_t<1024>, dim3(nv / 4096), dim3(1024), 0, 0, v));
    RUN("scramble x2049", hipLaunchKernelGGL(k_linear_scramble, dim3(nv / 1024), dim3(256), 0, 0, v, 2049u, (unsigned)(nv / 1024)));
    RUN("scramble x257", hipLaunchKernelGGL(k_linear_scramble, dim3(nv / 1024), dim3(256), 0, 0, v, 257u, (unsigned)(nv / 1024)));
    RUN("scramble x9", hipLaunchKernelGGL(k_linear_scramble, dim3(nv / 1024), dim3(256), 0, 0, v, 9u, (unsigned)(nv / 1024)));
    RUN("2 stores far", hipLaunchKernelGGL(k_linear_2far, dim3(nv / 2048), dim3(256), 0, 0, v, nv / 2));
    RUN("2 stores +1MiB", hipLaunchKernelGGL(k_linear_2far, dim3(nv / 2048), dim3(256), 0, 0, v, (size_t)262144));
    RUN("xstride 1 store", hipLaunchKernelGGL(k_xstride1, dim3(2, N, N / 4), dim3(256), 0, 0, v));
    for (unsigned sh = 0; sh < 2; sh++) {
        char nm[64]; snprintf(nm, sizeof nm, "linear 4K shift %u", sh);
        RUN(nm, hipLaunchKernelGGL(k_linear_shift<1>, dim3(nv / 1024), dim3(256), 0, 0, v, sh, (unsigned)(nv / 1024)));
    }
    RUN("linear 8K", hipLaunchKernelGGL(k_linear_shift<2>, dim3(nv / 2048), dim3(256), 0, 0, v, 0u, (unsigned)(nv / 2048)));
    RUN("linear 8K shift1", hipLaunchKernelGGL(k_linear_shift<2>, dim3(nv / 2048), dim3(256), 0, 0, v, 1u, (unsigned)(nv / 2048)));
    RUN("linear 16K", hipLaunchKernelGGL(k_linear_shift<4>, dim3(nv / 4096), dim3(256), 0, 0, v, 0u, (unsigned)(nv / 4096)));
    RUN("linear 32K", hipLaunchKernelGGL(k_linear_shift<8>, dim3(nv / 8192), dim3(256), 0, 0, v, 0u, (unsigned)(nv / 8192)));
    RUN("linear 32K shift1", hipLaunchKernelGGL(k_linear_shift<8>, dim3(nv / 8192), dim3(256), 0, 0, v, 1u, (unsigned)(nv / 8192)));
    RUN("memset", hipMemsetAsync(v, 0, nv * 4, 0));
    return 0;
}
