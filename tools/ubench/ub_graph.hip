// Micro-benchmark: host cost of queueing a chain of 9 small dependent kernels as 9 launches vs as ONE captured
// hipGraph launch (what a launch graph could save per sdfk_sample_march job on launch-bound grids).  Experiment
// harness, not part of the product.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 ub_graph.hip -o ub_graph
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k(float* p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.0f; }
int main()
{
    float* d; const int n = 1 << 16;
    CK(hipMalloc(&d, n * 4)); CK(hipMemset(d, 0, n * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto chain = [&] { for (int i = 0; i < 9; i++) hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, s, d, n); };
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); chain(); CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    const int R = 2000;
    for (int mode = 0; mode < 2; mode++) {
        for (int w = 0; w < 50; w++) { if (mode) CK(hipGraphLaunch(ge, s)); else chain(); }
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < R; r++) { if (mode) CK(hipGraphLaunch(ge, s)); else chain(); }
        auto t1 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(s));
        auto t2 = std::chrono::steady_clock::now();
        printf("%s: host %.2f us per chain to queue, %.2f us per chain end to end\n", mode ? "hipGraphLaunch (9 kernel nodes)" : "9 kernel launches          ",
               std::chrono::duration<double, std::micro>(t1 - t0).count() / R, std::chrono::duration<double, std::micro>(t2 - t0).count() / R);
    }
    return 0;
}
