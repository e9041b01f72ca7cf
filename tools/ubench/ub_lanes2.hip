// Micro-benchmark (sibling of ub_lanes.hip): WHICH streams are busy at the same time matters, not only how many.  Eight streams
// are created and touched in order (the runtime creates a stream's hardware queue at its first use); chains of 9 small dependent
// kernels (one captured graph launch per chain) are then dealt round-robin over a SUBSET of them, given as digits on the command
// line: ./ub_lanes2 0123 0124 04 0145 ...   Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 ub_lanes2.hip -o ub_lanes2
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
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
    const int wgs = 512, spin = 500, NS = 8;
    float* d[NS]; hipStream_t s[NS]; hipGraphExec_t ge[NS];
    for (int q = 0; q < NS; q++) {
        CK(hipMalloc(&d[q], wgs * 256 * 4)); CK(hipMemset(d[q], 0, wgs * 256 * 4));
        CK(hipStreamCreateWithFlags(&s[q], hipStreamNonBlocking));
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, s[q], d[q], 1);   // (first use: the hardware queue exists from here on)
        CK(hipStreamSynchronize(s[q]));
    }
    for (int q = 0; q < NS; q++) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(s[q], hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 9; i++) hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, s[q], d[q], spin);
        CK(hipStreamEndCapture(s[q], &g));
        CK(hipGraphInstantiate(&ge[q], g, nullptr, nullptr, 0));
    }
    const int R = 600;
    for (int a = 1; a < argc; a++) {
        const int K = (int)strlen(argv[a]);
        int idx[16];
        for (int j = 0; j < K; j++) idx[j] = (argv[a][j] - '0') % NS;
        for (int w = 0; w < 4 * K; w++) CK(hipGraphLaunch(ge[idx[w % K]], s[idx[w % K]]));
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < R; r++) CK(hipGraphLaunch(ge[idx[r % K]], s[idx[r % K]]));
        CK(hipDeviceSynchronize());
        auto t2 = std::chrono::steady_clock::now();
        printf("streams %-8s: %.2f us per chain\n", argv[a], std::chrono::duration<double, std::micro>(t2 - t0).count() / R);
    }
    return 0;
}
