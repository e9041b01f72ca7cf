// Micro-benchmark (round 6): can a COLOUR volume (values [x][y][z] + colours [x][y][z][3], 16 B per voxel) be written at the rate of
// a plain fill?  Round 2 (ub_colstore.hip) found that the rate falls with the number of 16-byte stores a lane issues: 6.7 TB/s with
// one, 5.9 with two, 5.4 with six -- and the sampler issues four per row (one value store, three colour stores), two rows per lane.
// New here: the MIXED store -- one voxel per lane, ONE 16-byte store per lane whose address is per lane: lanes 0..15 of a wavefront
// write the 64 voxels' values (256 B contiguous), lanes 16..63 their colours (768 B contiguous).  One store per lane, full 16-byte
// stores, both runs line-aligned.  Experiment harness, not part of the product.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 ub_mixstore.hip -o ub_mixstore
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int N = 512;
typedef float vf4 __attribute__((ext_vector_type(4)));

template <int F> __device__ __forceinline__ void st(float* p, vf4 v)
{
    if (F == 0) __builtin_nontemporal_store(v, reinterpret_cast<vf4*>(p));
    else if (F == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
    else *reinterpret_cast<vf4*>(p) = v;
}

// plain fill, S consecutive-KiB stores per lane
template <int S, int F>
__global__ __launch_bounds__(256) void k_fill(float* c)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* p = c + ((size_t)blockIdx.x * 4 + wave) * 256 * S;
    for (int q = 0; q < S; q++) st<F>(p + 256 * q + 4 * lane, vf4{(float)q, 2.f, 3.f, 4.f});
}

// MIXED: a wavefront = 64 consecutive voxels, one store per lane.  T = threads per workgroup; the voxel runs of a workgroup's wavefronts
// are consecutive (linear mapping)
template <int T, int F>
__global__ __launch_bounds__(T) void k_mix1(float* v, float* c)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t vox0 = ((size_t)blockIdx.x * (T / 64) + wave) * 64;
    float* p = lane < 16 ? v + vox0 + 4 * lane : c + vox0 * 3 + 4 * (lane - 16);
    st<F>(p, vf4{(float)lane, 2.f, 3.f, 4.f});
}
// MIXED, the sampler's tile: workgroup = 8 x rows x (T / 8) z of one y, wavefront = 64 z of one row (what the sign bytes need: 8 x per byte)
template <int T, int F>
__global__ __launch_bounds__(T) void k_mix1_tile(float* v, float* c)
{
    constexpr int ZW = T / 8 / 64;       // wavefronts along z per row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = wave / ZW, zc = wave % ZW;
    const int ix = blockIdx.z * 8 + r;
    const size_t vox0 = ((size_t)ix * N + blockIdx.y) * N + (size_t)blockIdx.x * (T / 8) + zc * 64;
    float* p = lane < 16 ? v + vox0 + 4 * lane : c + vox0 * 3 + 4 * (lane - 16);
    st<F>(p, vf4{(float)lane, 2.f, 3.f, 4.f});
}
// MIXED, R stores per lane: a wavefront covers R runs of 64 voxels (consecutive)
template <int R, int F>
__global__ __launch_bounds__(256) void k_mixR(float* v, float* c)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = 0; q < R; q++) {
        const size_t vox0 = (((size_t)blockIdx.x * 4 + wave) * R + q) * 64;
        float* p = lane < 16 ? v + vox0 + 4 * lane : c + vox0 * 3 + 4 * (lane - 16);
        st<F>(p, vf4{(float)lane, 2.f, 3.f, (float)q});
    }
}
// MIXED over 256 voxels with 4 stores per lane, but each store a FULL KiB of one array (value KiB, 3 colour KiB): the sampler today, 1 row / wavefront
template <int F>
__global__ __launch_bounds__(256) void k_vc4(float* v, float* c)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t w = (size_t)blockIdx.x * 4 + wave;
    st<F>(v + w * 256 + 4 * lane, vf4{1.f, 2.f, 3.f, 4.f});
    for (int q = 0; q < 3; q++) st<F>(c + w * 768 + 256 * q + 4 * lane, vf4{(float)q, 2.f, 3.f, 4.f});
}
// the sampler's tile today: 8 x rows x 256 z, RPW rows per wavefront, value store + three colour stores per row
template <int RPW, int F>
__global__ __launch_bounds__(512 / RPW) void k_tile_vc(float* v, float* c)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int rr = 0; rr < RPW; rr++) {
        const int ix = blockIdx.z * 8 + wave * RPW + rr;
        const size_t o = ((size_t)ix * N + blockIdx.y) * N + blockIdx.x * 256;
        st<F>(v + o + 4 * lane, vf4{1.f, 2.f, 3.f, 4.f});
        for (int q = 0; q < 3; q++) st<F>(c + o * 3 + 256 * q + 4 * lane, vf4{(float)q, 2.f, 3.f, 4.f});
    }
}
// the sampler's tile with the x rows' planes PADDED: row ix starts at ix * (N * N + pad) voxels -- does the power-of-two stride between
// the 8 rows of a workgroup (1 MiB for the values, 3 MiB for the colours at 512^3: the same channel for all of them?) cost the rate?
template <int RPW, int F>
__global__ __launch_bounds__(512 / RPW) void k_tile_vc_pad(float* v, float* c, size_t pad)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int rr = 0; rr < RPW; rr++) {
        const int ix = blockIdx.z * 8 + wave * RPW + rr;
        const size_t o = (size_t)ix * ((size_t)N * N + pad) + (size_t)blockIdx.y * N + blockIdx.x * 256;
        st<F>(v + o + 4 * lane, vf4{1.f, 2.f, 3.f, 4.f});
        for (int q = 0; q < 3; q++) st<F>(c + o * 3 + 256 * q + 4 * lane, vf4{(float)q, 2.f, 3.f, 4.f});
    }
}
// values only, the tile, padded planes
template <int F>
__global__ __launch_bounds__(256) void k_tile_v_pad(float* v, size_t pad)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int rr = 0; rr < 2; rr++) {
        const int ix = blockIdx.z * 8 + wave * 2 + rr;
        const size_t o = (size_t)ix * ((size_t)N * N + pad) + (size_t)blockIdx.y * N + blockIdx.x * 256;
        st<F>(v + o + 4 * lane, vf4{1.f, 2.f, 3.f, 4.f});
    }
}
// the tile with LONG runs: a workgroup = 8 x rows x (256 * ZL) z; per row ZL KiB of values and 3 ZL KiB of colours, contiguous
template <int ZL, int F>
__global__ __launch_bounds__(256) void k_tile_vc_long(float* v, float* c)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // blockIdx.x: run of 256 * ZL voxels within the (y, z) plane of the rows (N * N / 256 / ZL of them), blockIdx.z: x / 8
    for (int zi = 0; zi < ZL; zi++)
        for (int rr = 0; rr < 2; rr++) {
            const int ix = blockIdx.z * 8 + wave * 2 + rr;
            const size_t o = (size_t)ix * N * N + ((size_t)blockIdx.x * ZL + zi) * 256;
            st<F>(v + o + 4 * lane, vf4{1.f, 2.f, 3.f, 4.f});
            for (int q = 0; q < 3; q++) st<F>(c + o * 3 + 256 * q + 4 * lane, vf4{(float)q, 2.f, 3.f, 4.f});
        }
}
// MIXED store in the tile with padded planes (1024 threads: 8 rows x 128 z)
template <int F>
__global__ __launch_bounds__(1024) void k_mix1_tile_pad(float* v, float* c, size_t pad)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = wave / 2, zc = wave % 2;
    const int ix = blockIdx.z * 8 + r;
    const size_t vox0 = (size_t)ix * ((size_t)N * N + pad) + (size_t)blockIdx.y * N + (size_t)blockIdx.x * 128 + zc * 64;
    float* p = lane < 16 ? v + vox0 + 4 * lane : c + vox0 * 3 + 4 * (lane - 16);
    st<F>(p, vf4{(float)lane, 2.f, 3.f, 4.f});
}

// COLOURS ONLY, one voxel per lane: a wavefront = 64 consecutive voxels = 768 B of colours = lanes 0..47 store 16 B each (T threads / workgroup)
template <int T, int F>
__global__ __launch_bounds__(T) void k_c48(float* c)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t vox0 = ((size_t)blockIdx.x * (T / 64) + wave) * 64;
    if (lane < 48) st<F>(c + vox0 * 3 + 4 * lane, vf4{(float)lane, 2.f, 3.f, 4.f});
}
// COLOURS ONLY, four voxels per lane: a wavefront = 256 voxels = three full-KiB stores per lane (the colour part of the sampler today), linear
template <int F>
__global__ __launch_bounds__(256) void k_c3k(float* c)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t w = (size_t)blockIdx.x * 4 + wave;
    for (int q = 0; q < 3; q++) st<F>(c + w * 768 + 256 * q + 4 * lane, vf4{(float)q, 2.f, 3.f, 4.f});
}

// two-kernel split: values by one launch, colours by another (each one store per lane) -- what "a second pass" would cost in stores alone
template <int F>
__global__ __launch_bounds__(256) void k_one(float* a)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    st<F>(a + i, vf4{1.f, 2.f, 3.f, 4.f});
}

template <class L>
int timeit(const char* name, L launch, size_t bytes)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 100; i++) launch(i);   // (clocks)
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int R = 40;
    for (int i = 0; i < R; i++) launch(i);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-86s %7.1f us  %6.2f TB/s  %.3f of 8\n", name, ms / R * 1e3, bytes / (ms / R * 1e-3) / 1e12, bytes / (ms / R * 1e-3) / 8e12);
    return 0;
}

int main()
{
    const size_t nvox = (size_t)N * N * N;
    float *v[2], *c[2];
    const size_t maxpad = 8192;   // voxels
    for (auto& p : v) CK(hipMalloc(&p, (nvox + N * maxpad) * 4));
    for (auto& p : c) CK(hipMalloc(&p, (nvox + N * maxpad) * 12));
    const size_t B = nvox * 16;
    for (int rep = 0; rep < 2; rep++) {
        timeit("plain fill of the colour array (1.61 GB), 1 store / lane, nt", [&](int i) { hipLaunchKernelGGL((k_fill<1, 0>), dim3(nvox * 12 / 4096), dim3(256), 0, 0, c[i & 1]); }, nvox * 12);
        timeit("values then colours as TWO launches, 1 store / lane each, nt", [&](int i) { hipLaunchKernelGGL(k_one<0>, dim3(nvox / 1024), dim3(256), 0, 0, v[i & 1]); hipLaunchKernelGGL(k_one<0>, dim3(nvox * 3 / 1024), dim3(256), 0, 0, c[i & 1]); }, B);
        timeit("sampler today: tile, value + 3 colour KiB stores per row, 2 rows / wavefront, nt", [&](int i) { hipLaunchKernelGGL((k_tile_vc<2, 0>), dim3(N / 256, N, N / 8), dim3(256), 0, 0, v[i & 1], c[i & 1]); }, B);
        timeit("  the same, 1 row / wavefront (512 threads), nt", [&](int i) { hipLaunchKernelGGL((k_tile_vc<1, 0>), dim3(N / 256, N, N / 8), dim3(512), 0, 0, v[i & 1], c[i & 1]); }, B);
        timeit("  the same, 2 rows / wavefront, sc1 nt", [&](int i) { hipLaunchKernelGGL((k_tile_vc<2, 1>), dim3(N / 256, N, N / 8), dim3(256), 0, 0, v[i & 1], c[i & 1]); }, B);
        timeit("linear, value + 3 colour KiB stores (4 stores / lane), nt", [&](int i) { hipLaunchKernelGGL(k_vc4<0>, dim3(nvox / 1024), dim3(256), 0, 0, v[i & 1], c[i & 1]); }, B);
        timeit("MIXED store, 1 voxel / lane, linear, 256 threads, nt", [&](int i) { hipLaunchKernelGGL((k_mix1<256, 0>), dim3(nvox / 256), dim3(256), 0, 0, v[i & 1], c[i & 1]); }, B);
        timeit("MIXED store, 1 voxel / lane, linear, 512 threads, nt", [&](int i) { hipLaunchKernelGGL((k_mix1<512, 0>), dim3(nvox / 512), dim3(512), 0, 0, v[i & 1], c[i & 1]); }, B);
        timeit("MIXED store, 1 voxel / lane, linear, 1024 threads, nt", [&](int i) { hipLaunchKernelGGL((k_mix1<1024, 0>), dim3(nvox / 1024), dim3(1024), 0, 0, v[i & 1], c[i & 1]); }, B);
        timeit("MIXED store, 1 voxel / lane, linear, 256 threads, sc1 nt", [&](int i) { hipLaunchKernelGGL((k_mix1<256, 1>), dim3(nvox / 256), dim3(256), 0, 0, v[i & 1], c[i & 1]); }, B);
        timeit("MIXED store, 1 voxel / lane, linear, 256 threads, plain", [&](int i) { hipLaunchKernelGGL((k_mix1<256, 2>), dim3(nvox / 256), dim3(256), 0, 0, v[i & 1], c[i & 1]); }, B);
        timeit("MIXED store, tile 8 x rows x 64 z (512 threads), nt", [&](int i) { hipLaunchKernelGGL((k_mix1_tile<512, 0>), dim3(N / 64, N, N / 8), dim3(512), 0, 0, v[i & 1], c[i & 1]); }, B);
        timeit("MIXED store, tile 8 x rows x 128 z (1024 threads), nt", [&](int i) { hipLaunchKernelGGL((k_mix1_tile<1024, 0>), dim3(N / 128, N, N / 8), dim3(1024), 0, 0, v[i & 1], c[i & 1]); }, B);
        timeit("MIXED store, tile 8 x rows x 128 z (1024 threads), sc1 nt", [&](int i) { hipLaunchKernelGGL((k_mix1_tile<1024, 1>), dim3(N / 128, N, N / 8), dim3(1024), 0, 0, v[i & 1], c[i & 1]); }, B);
        timeit("COLOURS ONLY, 1 voxel / lane, lanes 0..47 store (256 threads), nt", [&](int i) { hipLaunchKernelGGL((k_c48<256, 0>), dim3(nvox / 256), dim3(256), 0, 0, c[i & 1]); }, nvox * 12);
        timeit("COLOURS ONLY, 1 voxel / lane, lanes 0..47 store (1024 threads), nt", [&](int i) { hipLaunchKernelGGL((k_c48<1024, 0>), dim3(nvox / 1024), dim3(1024), 0, 0, c[i & 1]); }, nvox * 12);
        timeit("COLOURS ONLY, 1 voxel / lane, lanes 0..47 store (256 threads), sc1 nt", [&](int i) { hipLaunchKernelGGL((k_c48<256, 1>), dim3(nvox / 256), dim3(256), 0, 0, c[i & 1]); }, nvox * 12);
        timeit("COLOURS ONLY, 4 voxels / lane, 3 full-KiB stores / lane, linear, nt", [&](int i) { hipLaunchKernelGGL((k_c3k<0>), dim3(nvox / 1024), dim3(256), 0, 0, c[i & 1]); }, nvox * 12);
        timeit("COLOURS ONLY, 4 voxels / lane, 3 full-KiB stores / lane, linear, sc1 nt", [&](int i) { hipLaunchKernelGGL((k_c3k<1>), dim3(nvox / 1024), dim3(256), 0, 0, c[i & 1]); }, nvox * 12);
        if (rep == 0)
        for (size_t pad : {(size_t)0, (size_t)64, (size_t)1024, (size_t)1088, (size_t)4160}) {
            char nm[160];
            snprintf(nm, sizeof nm, "sampler tile, x planes padded by %zu voxels, 2 rows / wavefront, nt", pad);
            timeit(nm, [&](int i) { hipLaunchKernelGGL((k_tile_vc_pad<2, 0>), dim3(N / 256, N, N / 8), dim3(256), 0, 0, v[i & 1], c[i & 1], pad); }, B);
            snprintf(nm, sizeof nm, "  the same, sc1 nt (pad %zu)", pad);
            timeit(nm, [&](int i) { hipLaunchKernelGGL((k_tile_vc_pad<2, 1>), dim3(N / 256, N, N / 8), dim3(256), 0, 0, v[i & 1], c[i & 1], pad); }, B);
            snprintf(nm, sizeof nm, "  MIXED tile 8 x 128 z, sc1 nt (pad %zu)", pad);
            timeit(nm, [&](int i) { hipLaunchKernelGGL((k_mix1_tile_pad<1>), dim3(N / 128, N, N / 8), dim3(1024), 0, 0, v[i & 1], c[i & 1], pad); }, B);
            snprintf(nm, sizeof nm, "  VALUES ONLY tile, nt (pad %zu)", pad);
            timeit(nm, [&](int i) { hipLaunchKernelGGL((k_tile_v_pad<0>), dim3(N / 256, N, N / 8), dim3(256), 0, 0, v[i & 1], pad); }, nvox * 4);
        }
        timeit("VALUES ONLY linear fill, 1 store / lane, nt", [&](int i) { hipLaunchKernelGGL((k_fill<1, 0>), dim3(nvox * 4 / 4096), dim3(256), 0, 0, v[i & 1]); }, nvox * 4);
        timeit("sampler tile with runs of 1024 voxels per row and workgroup, nt", [&](int i) { hipLaunchKernelGGL((k_tile_vc_long<4, 0>), dim3(N * N / 256 / 4, 1, N / 8), dim3(256), 0, 0, v[i & 1], c[i & 1]); }, B);
        timeit("sampler tile with runs of 4096 voxels per row and workgroup, nt", [&](int i) { hipLaunchKernelGGL((k_tile_vc_long<16, 0>), dim3(N * N / 256 / 16, 1, N / 8), dim3(256), 0, 0, v[i & 1], c[i & 1]); }, B);
        timeit("sampler tile with runs of 4096 voxels per row and workgroup, sc1 nt", [&](int i) { hipLaunchKernelGGL((k_tile_vc_long<16, 1>), dim3(N * N / 256 / 16, 1, N / 8), dim3(256), 0, 0, v[i & 1], c[i & 1]); }, B);
        timeit("MIXED store, 2 runs of 64 voxels per wavefront (2 stores / lane), nt", [&](int i) { hipLaunchKernelGGL((k_mixR<2, 0>), dim3(nvox / 512), dim3(256), 0, 0, v[i & 1], c[i & 1]); }, B);
        timeit("MIXED store, 4 runs of 64 voxels per wavefront (4 stores / lane), nt", [&](int i) { hipLaunchKernelGGL((k_mixR<4, 0>), dim3(nvox / 1024), dim3(256), 0, 0, v[i & 1], c[i & 1]); }, B);
    }
    return 0;
}
