// reproduction: rs < ws ? 0 : min(rs - ws, wc + 1) + 1 in a strided LDS fill loop (k_vertices' row starts relative to their window)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
__global__ void f(const uint32_t* rowstart, uint32_t* out, uint32_t n, int nrs, uint32_t w1s, uint32_t w2s, uint32_t w1c, uint32_t w2c)
{
    __shared__ uint32_t s[2][512];
    for (int i = threadIdx.x; i < 2 * nrs; i += 256) {
        const int w = i >= nrs, k = w ? i - nrs : i;
        const uint32_t rs = min(rowstart[i], n);
        const uint32_t ws = w ? w2s : w1s, wc = w ? w2c : w1c;
#if VARIANT == 0
        s[w][k] = (uint32_t)(rs < ws ? 0u : min(rs - ws, wc + 1u) + 1u);
#elif VARIANT == 1
        const long long d = (long long)rs + 1 - (long long)ws;
        s[w][k] = (uint32_t)max(0ll, min(d, (long long)wc + 2));
#else
        uint32_t e = 0;
        if (rs >= ws) { e = rs - ws; e = e > wc + 1u ? wc + 1u : e; e += 1u; }
        s[w][k] = e;
#endif
    }
    __syncthreads();
    out[threadIdx.x] = s[threadIdx.x >> 7][threadIdx.x & 127];
}
int main()
{
    uint32_t h[64], *d, *o, r[256];
    for (int i = 0; i < 64; i++) h[i] = 239 + 4 * i;
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof r);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(f, dim3(1), dim3(256), 0, 0, d, o, 314u, 21, 240u, 281u, 74u, 33u);
    hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    printf("s[0][0..3] = %u %u %u %u (expect 0 4 8 12)   s[1][0..2] = %u %u %u\n", r[0], r[1], r[2], r[3], r[128], r[129], r[130]);
    return 0;
}
