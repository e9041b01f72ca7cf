// Micro-benchmark: does hipExtStreamCreateWithCUMask restrict a kernel on this box, and what does a
// write-bound fill kernel lose on a subset of the CUs?   hipcc --offload-arch=gfx950 -O3 ub_cumask.hip -o ub_cumask.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_fill(f4v* p, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(f4v{1.f, 2.f, 3.f, 4.f}, p + i);
}
__global__ __launch_bounds__(256) void k_spin(float* p, int iters)
{
    float a = threadIdx.x;
    for (int i = 0; i < iters; i++) a = a * 1.0001f + 0.5f;
    if (a == 12345.f) p[0] = a;
}
static float time_on(hipStream_t s, void (*launch)(hipStream_t))
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(s); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int k = 0; k < 5; k++) launch(s);
    CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 5 * 1e3f;
}
static f4v* g_buf; static size_t g_n;
static void l_fill(hipStream_t s) { hipLaunchKernelGGL(k_fill, dim3((unsigned)(g_n / 256)), dim3(256), 0, s, g_buf, g_n); }
static void l_spin(hipStream_t s) { hipLaunchKernelGGL(k_spin, dim3(256 * 8), dim3(256), 0, s, (float*)g_buf, 20000); }
int main()
{
    g_n = (512u << 20) / 16;
    CK(hipMalloc(&g_buf, g_n * 16));
    const uint32_t pats[] = {0xffffffffu, 0x77777777u, 0x55555555u, 0x11111111u, 0x0000ffffu, 0x000000ffu};
    for (uint32_t w : pats) {
        uint32_t mask[8]; for (int k = 0; k < 8; k++) mask[k] = w;
        hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, 8, mask));
        printf("mask %08x x8: fill 512 MiB %8.1f us   spin %8.1f us\n", w, time_on(s, l_fill), time_on(s, l_spin));
        CK(hipStreamDestroy(s));
    }
    uint32_t one[8] = {0xffffffffu, 0, 0, 0, 0, 0, 0, 0};
    hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, 8, one));
    printf("mask word0 only: fill %8.1f us   spin %8.1f us\n", time_on(s, l_fill), time_on(s, l_spin));
    return 0;
}
