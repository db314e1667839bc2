// global atomic throughput on gfx950: n atomicAdd's over m counters (uniform pseudo-random addresses), returning / non-returning,
// one per thread.   hipcc --offload-arch=gfx950 -O3 scripts/ubench/atomics.hip -o scripts/ubench/atomics && scripts/ubench/atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <bool RET, int PER>
__global__ void k(uint32_t *ctr, uint32_t m, uint32_t n, uint32_t *sink, int skew)
{
    const uint32_t i0 = (blockIdx.x * blockDim.x + threadIdx.x) * PER;
    uint32_t acc = 0;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const uint32_t i = i0 + u;
        if (i >= n) break;
        uint32_t h = hash(i);
        uint32_t a = h % m;
        if (skew) { const uint32_t r = (h >> 8) & 255u; if (r < 128u) a = a % (m / 64u); }   // half of the traffic on 1/64 of the counters
        if (RET) acc += atomicAdd(&ctr[a], 1u);
        else atomicAdd(&ctr[a], 1u);
    }
    if (RET && acc == 0xFFFFFFFFu) sink[0] = acc;
}

int main()
{
    const uint32_t n = 4840000;
    uint32_t *ctr, *sink;
    CK(hipMalloc(&ctr, 4 * (1 << 20)));
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (uint32_t m : {1024u, 32768u, 1u << 20}) {
        for (int skew = 0; skew < 2; ++skew)
            for (int ret = 0; ret < 2; ++ret) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    CK(hipMemset(ctr, 0, 4 * (1 << 20)));
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0));
                    const int PER = 4;
                    const unsigned grid = (n / PER + 255) / 256;
                    if (ret) k<true, PER><<<grid, 256>>>(ctr, m, n, sink, skew);
                    else k<false, PER><<<grid, 256>>>(ctr, m, n, sink, skew);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms < best ? ms : best;
                }
                printf("counters %8u  skew %d  %s: %8.1f us  = %.2f G atomics/s\n", m, skew, ret ? "returning    " : "non-returning", best * 1e3, n / best * 1e-6);
            }
    }
    return 0;
}
