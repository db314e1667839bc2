// Micro-benchmark of the render-backward inner loop (lane = instance, 256 pixels per instance).
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize bwd_variants.hip -o bwd_variants
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int VAR>
__global__ void __launch_bounds__(256) k(const float4 *__restrict__ rec, const float *__restrict__ dL, float *__restrict__ out, int R, int W)
{
    __shared__ float4 gtile[4][64];
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (k >= R) return;
    const float4 a = rec[2 * k], b = rec[2 * k + 1];
    float4 *gt = gtile[wave];
    const int t = __builtin_amdgcn_readfirstlane(k >> 6) & 1023;
    const int x0 = (t & 31) * 16, y0 = (t >> 5) * 16;
    {
        const int ry = y0 + (lane >> 2), cx = x0 + (lane & 3) * 4;
        gt[lane] = *reinterpret_cast<const float4 *>(dL + (size_t)ry * W + cx);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    float S0 = 0, S1 = 0, S2 = 0, S3 = 0, S4 = 0, S5 = 0;
    const float dx0 = a.x - (float)x0;
    constexpr int UR = (VAR == 1) ? 1 : 2;
#pragma unroll UR
    for (int r = 0; r < 16; ++r) {
        const float dy = a.y - (float)(y0 + r);
        const float bdy = a.w * dy;
        const float cdy2 = (b.x * dy) * dy;
        float g[16];
        if (VAR == 5) {
            const float *row = dL + (size_t)(y0 + r) * W + x0;   // uniform -> scalar loads
#pragma unroll
            for (int c = 0; c < 16; ++c) g[c] = row[c];
        } else {
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const float4 v = gt[r * 4 + c4];
                g[4 * c4] = v.x; g[4 * c4 + 1] = v.y; g[4 * c4 + 2] = v.z; g[4 * c4 + 3] = v.w;
            }
        }
        float r0 = 0, r1 = 0, r3 = 0;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float dx = dx0 - (float)c;
            const float p2 = dx * (a.z * dx + bdy) + cdy2;
            float G;
            if (VAR == 3) G = p2; else G = __builtin_amdgcn_exp2f(p2);
            float w;
            if (VAR == 2) w = G * g[c];
            else if (VAR == 6) w = fminf(fmaxf(-p2 * 1e30f, 0.f), 1.f) * (G * g[c]);   // branch-free mask
            else {
                const bool ok = (p2 <= 0.0f) && (b.y * G >= 0.00001f);
                w = ok ? G * g[c] : 0.f;
            }
            const float wdx = w * dx;
            r0 += w; r1 += wdx; r3 += wdx * dx;
        }
        S0 += r0; S1 += r1; S3 += r3; S2 += dy * r0; S4 += dy * r1; S5 += dy * dy * r0;
    }
    out[k] = S0 + S1 + S2 + S3 + S4 + S5;
}

// forward-like: lane = pixel, Gaussians broadcast from LDS
template <int VAR>
__global__ void __launch_bounds__(256) kf(const float4 *__restrict__ rec, float *__restrict__ out, int L)
{
    __shared__ float4 sA[256];
    __shared__ float2 sB[256];
    const int tid = threadIdx.x;
    const float fx = (float)(tid & 15), fy = (float)(tid >> 4);
    float C = 0.f;
    for (int base = 0; base < L; base += 256) {
        __syncthreads();
        const int id = (blockIdx.x * 131 + base + tid) & 0xFFFF;
        const float4 a = rec[2 * id], b = rec[2 * id + 1];
        sA[tid] = a; sB[tid] = make_float2(b.x, b.y);
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < 256; ++j) {
            const float4 a = sA[j]; const float2 b = sB[j];
            const float dx = a.x - fx, dy = a.y - fy;
            const float p2 = dx * (a.z * dx + a.w * dy) + (b.x * dy) * dy;
            const float alpha = b.y * __builtin_amdgcn_exp2f(p2);
            const bool ok = (p2 <= 0.0f) && (alpha >= 0.00001f);
            C += ok ? alpha : 0.f;
        }
    }
    out[blockIdx.x * 256 + tid] = C;
}

template <int VAR>
float run(const float4 *rec, const float *dL, float *out, int R, int W)
{
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) k<VAR><<<(R + 255) / 256, 256>>>(rec, dL, out, R, W);
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) k<VAR><<<(R + 255) / 256, 256>>>(rec, dL, out, R, W);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / 10 * 1e3f;
}

int main()
{
    const int R = 1156000, W = 512;
    std::vector<float> h(8 * (size_t)R);
    for (int i = 0; i < R; ++i) {
        float *r = &h[8 * (size_t)i];
        const int t = (i >> 6) & 1023;
        r[0] = (t & 31) * 16 + (rand() % 1600) / 100.f; r[1] = (t >> 5) * 16 + (rand() % 1600) / 100.f;
        r[2] = -0.05f; r[3] = 0.01f; r[4] = -0.04f; r[5] = 0.01f; r[6] = 0.1f; r[7] = 0.1f;
    }
    std::vector<float> hd(W * W);
    for (auto &v : hd) v = (rand() % 2000 - 1000) / 1e6f;
    float4 *rec; float *dL, *out;
    CHECK(hipMalloc(&rec, h.size() * 4)); CHECK(hipMalloc(&dL, hd.size() * 4)); CHECK(hipMalloc(&out, (size_t)R * 4));
    CHECK(hipMemcpy(rec, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dL, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
    printf("bwd V0 base        : %8.1f us\n", run<0>(rec, dL, out, R, W));
    printf("bwd V1 unroll rows1: %8.1f us\n", run<1>(rec, dL, out, R, W));
    printf("bwd V2 no mask     : %8.1f us\n", run<2>(rec, dL, out, R, W));
    printf("bwd V3 no exp      : %8.1f us\n", run<3>(rec, dL, out, R, W));
    printf("bwd V5 sgpr dL     : %8.1f us\n", run<5>(rec, dL, out, R, W));
    printf("bwd V6 arith mask  : %8.1f us\n", run<6>(rec, dL, out, R, W));
    // forward-like: 1024 tiles x list length L
    for (int L : {1024, 4096}) {
        hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        kf<0><<<1024, 256>>>(rec, out, L);
        CHECK(hipEventRecord(a));
        for (int i = 0; i < 5; ++i) kf<0><<<1024, 256>>>(rec, out, L);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        printf("fwd uniform L=%d   : %8.1f us  (%.1f Gpairs/s)\n", L, ms / 5 * 1e3f, 1024.0 * 256 * L / (ms / 5 * 1e-3) / 1e9);
    }
    printf("bwd pairs/s at V0: R*256\n");
    return 0;
}
