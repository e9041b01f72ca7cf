// Micro-benchmark: store patterns of the COLOUR volume (512^3 x rgb floats = 1.61 GB, layout [x][y][z][3]) as the fused
// sampler writes it -- 8 x rows per workgroup, a wavefront stores 3 KiB contiguous per row as three 1 KiB nontemporal
// instructions -- against a plain linear fill of the same bytes with the same instructions, and a few re-mappings.
// Experiment harness, not part of the product.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 ub_colstore.hip -o ub_colstore
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int N = 512;
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st(float* p, float a) { vf4 t = {a, 2.f, 3.f, 4.f}; __builtin_nontemporal_store(t, reinterpret_cast<vf4*>(p)); }

// V1: the sampler's pattern.  grid (N/256, N, N/8), 256 threads: wave w -> rows 2w, 2w+1 of the 8
template <int RPW>
__global__ __launch_bounds__(512 / RPW) void k_tile(float* c)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int rr = 0; rr < RPW; rr++) {
        const int ix = blockIdx.z * 8 + wave * RPW + rr;
        float* p = c + (((size_t)ix * N + blockIdx.y) * N + blockIdx.x * 256) * 3;
        for (int q = 0; q < 3; q++) st(p + 256 * q + 4 * lane, (float)q);
    }
}
// V2: plain linear fill with the same per-lane work (RPW * 3 stores of 16 B, each wavefront 3 KiB contiguous per "row")
template <int RPW>
__global__ __launch_bounds__(512 / RPW) void k_linear(float* c)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t wg = blockIdx.x;
    for (int rr = 0; rr < RPW; rr++) {
        float* p = c + ((wg * 8 + wave * RPW + rr) * 256) * 3;
        for (int q = 0; q < 3; q++) st(p + 256 * q + 4 * lane, (float)q);
    }
}
// V3: one x row per workgroup, 8 consecutive 256-voxel chunks of its (y,z) plane (24 KiB contiguous per workgroup)
template <int RPW>
__global__ __launch_bounds__(512 / RPW) void k_plane(float* c)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // blockIdx.x: chunk group within the plane (N*N/256/8 of them), blockIdx.y: x row
    for (int rr = 0; rr < RPW; rr++) {
        float* p = c + ((size_t)blockIdx.y * N * N + ((size_t)blockIdx.x * 8 + wave * RPW + rr) * 256) * 3;
        for (int q = 0; q < 3; q++) st(p + 256 * q + 4 * lane, (float)q);
    }
}
// V4: the tile, but the 8 rows of a workgroup are 8 consecutive y rows of ONE x (still 8 pieces of 3 KiB, 6 KiB apart)
template <int RPW>
__global__ __launch_bounds__(512 / RPW) void k_ytile(float* c)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int rr = 0; rr < RPW; rr++) {
        const int iy = blockIdx.y * 8 + wave * RPW + rr;
        float* p = c + (((size_t)blockIdx.z * N + iy) * N + blockIdx.x * 256) * 3;
        for (int q = 0; q < 3; q++) st(p + 256 * q + 4 * lane, (float)q);
    }
}

// V5: plain linear fill, S consecutive-KiB stores per lane (S = 1: the classic fill), `bytes` in total
template <int S>
__global__ __launch_bounds__(256) void k_fill(float* c)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* p = c + ((size_t)blockIdx.x * 4 + wave) * 256 * S;
    for (int q = 0; q < S; q++) st(p + 256 * q + 4 * lane, (float)q);
}

// V6: values + colours with ONE voxel per lane: a 4-byte and a 12-byte store per lane (256 B + 768 B contiguous per wavefront)
typedef float vf3 __attribute__((ext_vector_type(3)));
__global__ __launch_bounds__(256) void k_vc1(float* v, float* c)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    __builtin_nontemporal_store((float)i, v + i);
    vf3 t = {1.f, 2.f, (float)i};
    __builtin_nontemporal_store(t, reinterpret_cast<vf3*>(c + 3 * i));
}
// V7: values + colours, 4 voxels per lane, colours as three 1 KiB stores (the sampler: 4 stores per lane and row), 1 row / wavefront
__global__ __launch_bounds__(256) void k_vc4(float* v, float* c)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t w = (size_t)blockIdx.x * 4 + wave;
    st(v + w * 256 + 4 * lane, 1.f);
    for (int q = 0; q < 3; q++) st(c + w * 768 + 256 * q + 4 * lane, (float)q);
}
// V8: colours only, one voxel per lane (12-byte stores)
__global__ __launch_bounds__(256) void k_c1(float* c)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    vf3 t = {1.f, 2.f, (float)i};
    __builtin_nontemporal_store(t, reinterpret_cast<vf3*>(c + 3 * i));
}

// V9: the sampler's tile with BOTH arrays: per row one value store + three colour stores; ORDER: 0 = value first, 1 = colours first
template <int RPW, int ORDER>
__global__ __launch_bounds__(512 / RPW) void k_tile_vc(float* v, float* c)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int rr = 0; rr < RPW; rr++) {
        const int ix = blockIdx.z * 8 + wave * RPW + rr;
        const size_t o = ((size_t)ix * N + blockIdx.y) * N + blockIdx.x * 256;
        if (ORDER == 0) st(v + o + 4 * lane, 1.f);
        for (int q = 0; q < 3; q++) st(c + o * 3 + 256 * q + 4 * lane, (float)q);
        if (ORDER == 1) st(v + o + 4 * lane, 1.f);
    }
}
// V10: linear, both arrays, RPW rows per wavefront
template <int RPW>
__global__ __launch_bounds__(512 / RPW) void k_lin_vc(float* v, float* c)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int rr = 0; rr < RPW; rr++) {
        const size_t w = ((size_t)blockIdx.x * 8 + wave * RPW + rr);
        st(v + w * 256 + 4 * lane, 1.f);
        for (int q = 0; q < 3; q++) st(c + w * 768 + 256 * q + 4 * lane, (float)q);
    }
}

template <class F>
int timeit(const char* name, F launch, size_t bytes)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 200; i++) launch(i);   // (clocks)
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int R = 40;
    for (int i = 0; i < R; i++) launch(i);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-64s %7.1f us  %6.2f TB/s\n", name, ms / R * 1e3, bytes / (ms / R * 1e-3) / 1e12);
    return 0;
}

int main()
{
    const size_t nf = (size_t)N * N * N * 3;
    float* c[4];
    for (auto& p : c) CK(hipMalloc(&p, nf * 4));
    const size_t bytes = nf * 4;
    timeit("tile 8 x-rows x 256 z (sampler), 2 rows / wavefront", [&](int i) { hipLaunchKernelGGL(k_tile<2>, dim3(N / 256, N, N / 8), dim3(256), 0, 0, c[i & 3]); }, bytes);
    timeit("tile 8 x-rows x 256 z, 1 row / wavefront (512 threads)", [&](int i) { hipLaunchKernelGGL(k_tile<1>, dim3(N / 256, N, N / 8), dim3(512), 0, 0, c[i & 3]); }, bytes);
    timeit("linear, same instructions, 2 rows / wavefront", [&](int i) { hipLaunchKernelGGL(k_linear<2>, dim3(N * N * N / 256 / 8), dim3(256), 0, 0, c[i & 3]); }, bytes);
    timeit("linear, 1 row / wavefront (512 threads)", [&](int i) { hipLaunchKernelGGL(k_linear<1>, dim3(N * N * N / 256 / 8), dim3(512), 0, 0, c[i & 3]); }, bytes);
    timeit("one x row per workgroup: 24 KiB contiguous, 2 rows / wavefront", [&](int i) { hipLaunchKernelGGL(k_plane<2>, dim3(N * N / 256 / 8, N), dim3(256), 0, 0, c[i & 3]); }, bytes);
    timeit("tile of 8 y rows of one x, 2 rows / wavefront", [&](int i) { hipLaunchKernelGGL(k_ytile<2>, dim3(N / 256, N / 8, N), dim3(256), 0, 0, c[i & 3]); }, bytes);
    const unsigned nwave = (unsigned)(bytes / 1024);
    timeit("plain fill 1.61 GB, 1 store / lane", [&](int i) { hipLaunchKernelGGL(k_fill<1>, dim3(nwave / 4), dim3(256), 0, 0, c[i & 3]); }, bytes);
    timeit("plain fill 1.61 GB, 2 stores / lane", [&](int i) { hipLaunchKernelGGL(k_fill<2>, dim3(nwave / 8), dim3(256), 0, 0, c[i & 3]); }, bytes);
    timeit("plain fill 1.61 GB, 3 stores / lane", [&](int i) { hipLaunchKernelGGL(k_fill<3>, dim3(nwave / 12), dim3(256), 0, 0, c[i & 3]); }, bytes);
    timeit("plain fill 1.61 GB, 6 stores / lane", [&](int i) { hipLaunchKernelGGL(k_fill<6>, dim3(nwave / 24), dim3(256), 0, 0, c[i & 3]); }, bytes);
    const size_t small = (size_t)512 << 20;
    timeit("plain fill 0.54 GB (4 buffers in turn), 1 store / lane", [&](int i) { hipLaunchKernelGGL(k_fill<1>, dim3(small / 4096), dim3(256), 0, 0, c[i & 3]); }, small);
    timeit("plain fill 0.54 GB (4 buffers in turn), 3 stores / lane", [&](int i) { hipLaunchKernelGGL(k_fill<3>, dim3(small / 4096 / 3), dim3(256), 0, 0, c[i & 3]); }, small / 12288 * 12288);
    timeit("plain fill 0.54 GB (ONE buffer), 1 store / lane", [&](int i) { hipLaunchKernelGGL(k_fill<1>, dim3(small / 4096), dim3(256), 0, 0, c[0]); }, small);
    float* v[4];
    for (auto& p : v) CK(hipMalloc(&p, (size_t)N * N * N * 4));
    const size_t nvox = (size_t)N * N * N;
    timeit("values + colours (2.15 GB), 1 voxel / lane: 4 B + 12 B stores", [&](int i) { hipLaunchKernelGGL(k_vc1, dim3(nvox / 256), dim3(256), 0, 0, v[i & 3], c[i & 3]); }, nvox * 16);
    timeit("values + colours (2.15 GB), 4 voxels / lane: 16 B + 3 x 16 B stores", [&](int i) { hipLaunchKernelGGL(k_vc4, dim3(nvox / 1024), dim3(256), 0, 0, v[i & 3], c[i & 3]); }, nvox * 16);
    timeit("colours only (1.61 GB), 1 voxel / lane: one 12 B store", [&](int i) { hipLaunchKernelGGL(k_c1, dim3(nvox / 256), dim3(256), 0, 0, c[i & 3]); }, nvox * 12);
    timeit("tile, values + colours, 2 rows / wavefront (the sampler)", [&](int i) { hipLaunchKernelGGL((k_tile_vc<2, 0>), dim3(N / 256, N, N / 8), dim3(256), 0, 0, v[i & 3], c[i & 3]); }, nvox * 16);
    timeit("tile, values + colours, 1 row / wavefront", [&](int i) { hipLaunchKernelGGL((k_tile_vc<1, 0>), dim3(N / 256, N, N / 8), dim3(512), 0, 0, v[i & 3], c[i & 3]); }, nvox * 16);
    timeit("tile, colours then value, 2 rows / wavefront", [&](int i) { hipLaunchKernelGGL((k_tile_vc<2, 1>), dim3(N / 256, N, N / 8), dim3(256), 0, 0, v[i & 3], c[i & 3]); }, nvox * 16);
    timeit("linear, values + colours, 2 rows / wavefront", [&](int i) { hipLaunchKernelGGL(k_lin_vc<2>, dim3(nvox / 2048), dim3(256), 0, 0, v[i & 3], c[i & 3]); }, nvox * 16);
    timeit("linear, values + colours, 1 row / wavefront (512 threads)", [&](int i) { hipLaunchKernelGGL(k_lin_vc<1>, dim3(nvox / 2048), dim3(512), 0, 0, v[i & 3], c[i & 3]); }, nvox * 16);
    timeit("plain fill 1.61 GB, 3 stores / lane (again)", [&](int i) { hipLaunchKernelGGL(k_fill<3>, dim3(nwave / 12), dim3(256), 0, 0, c[i & 3]); }, bytes);
    timeit("plain fill 1.61 GB, 4 stores / lane", [&](int i) { hipLaunchKernelGGL(k_fill<4>, dim3(nwave / 16), dim3(256), 0, 0, c[i & 3]); }, bytes);
    return 0;
}
