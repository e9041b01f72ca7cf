// Micro-benchmark: throughput of chains of 9 small dependent kernels (one captured hipGraphLaunch per chain) issued round-robin
// over K streams, K = 1..6 -- how many independent chains does this part overlap?  (The library's sharded step gains up to three
// lanes and loses with a fourth: tools/slab_chain_probe.py.  Is that the hardware, or the library?)  Each kernel: `wgs` workgroups
// spinning ~`spin` clock ticks (a latency-bound kernel at low occupancy, like the meshing chain of a small slab).
// Experiment harness, not part of the product.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 ub_lanes.hip -o ub_lanes
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k(float* p, int spin)
{
    const long long t0 = wall_clock64();
    float a = p[blockIdx.x * 256 + threadIdx.x];
    while (wall_clock64() - t0 < spin) a = a * 1.0001f + 1.0f;
    p[blockIdx.x * 256 + threadIdx.x] = a;
}
int main(int argc, char** argv)
{
    const int wgs = argc > 1 ? atoi(argv[1]) : 512, spin = argc > 2 ? atoi(argv[2]) : 500;   // wall_clock64: 100 MHz -> 500 ticks = 5 us
    const int KMAX = 6;
    float* d[KMAX]; hipStream_t s[KMAX]; hipGraphExec_t ge[KMAX];
    for (int q = 0; q < KMAX; q++) {
        CK(hipMalloc(&d[q], wgs * 256 * 4)); CK(hipMemset(d[q], 0, wgs * 256 * 4));
        CK(hipStreamCreateWithFlags(&s[q], hipStreamNonBlocking));
        hipGraph_t g;
        CK(hipStreamBeginCapture(s[q], hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 9; i++) hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, s[q], d[q], spin);
        CK(hipStreamEndCapture(s[q], &g));
        CK(hipGraphInstantiate(&ge[q], g, nullptr, nullptr, 0));
    }
    const int R = 600;
    for (int K = 1; K <= KMAX; K++) {
        for (int w = 0; w < 4 * K; w++) CK(hipGraphLaunch(ge[w % K], s[w % K]));
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < R; r++) CK(hipGraphLaunch(ge[r % K], s[r % K]));
        auto t1 = std::chrono::steady_clock::now();
        CK(hipDeviceSynchronize());
        auto t2 = std::chrono::steady_clock::now();
        printf("%d streams: %.2f us per chain (host %.2f us to queue one)\n", K, std::chrono::duration<double, std::micro>(t2 - t0).count() / R,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / R);
    }
    return 0;
}
