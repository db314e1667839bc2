// ubench_clock.hip -- VALU issue cost on gfx950 in SHADER CYCLES, and the shader clock under load.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_clock.hip -o scripts/ubench_clock && scripts/ubench_clock
// Round 2's ubench_valu.hip turned the HIP-event wall time of a 30-100 us kernel into cycles at the NOMINAL 2.4 GHz; that
// cannot separate the issue rate from the real clock or from launch overhead.  Here every wave reads
// s_memtime (shader-clock ticks) and s_memrealtime (100 MHz constant clock) around >= 1 ms of work:
//   cycles per wave-instruction per SIMD = d(s_memtime) / (instructions the SIMD's resident waves issued)
//   shader clock under this load         = d(s_memtime) / d(s_memrealtime) * 100 MHz
// and HW_ID / XCC_ID say where the waves really ran (waves per SIMD as dispatched, not as hoped).
// Bodies: plain v_fma / v_mul / v_exp streams, and the two render inner loops exactly as the kernels issue them
// (backward: v_cmpx + mul + add + 2 fmac + s_mov exec + 2 recurrence muls per pixel; forward: cmp + cndmask + add + 2 muls).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Stamp { unsigned long long c0, c1, r0, r1; unsigned hw, xcc; };

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

__device__ __forceinline__ void stamp_begin(Stamp &s)
{
    s.r0 = __builtin_amdgcn_s_memrealtime();
    s.c0 = __builtin_amdgcn_s_memtime();
}
__device__ __forceinline__ void stamp_end(Stamp &s, Stamp *out)
{
    s.c1 = __builtin_amdgcn_s_memtime();
    s.r1 = __builtin_amdgcn_s_memrealtime();
    s.hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
    s.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;
}

// 8 independent accumulators, one instruction each per iteration
#define STREAM_KERNEL(NAME, ASM)                                                                                          \
    __global__ void NAME(float *out, Stamp *st, int iters, float a, float b)                                              \
    {                                                                                                                     \
        float acc[8];                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) acc[i] = (float)threadIdx.x * 1e-3f + i;                            \
        Stamp s;                                                                                                          \
        stamp_begin(s);                                                                                                   \
        for (int it = 0; it < iters; ++it) {                                                                              \
            REP8(ASM)                                                                                                     \
        }                                                                                                                 \
        stamp_end(s, st);                                                                                                 \
        float r = 0;                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) r += acc[i];                                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                                   \
    }
#define S_FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#define S_MUL(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(a));
#define S_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(acc[i]));
STREAM_KERNEL(k_fma, S_FMA)
STREAM_KERNEL(k_mul, S_MUL)
STREAM_KERNEL(k_exp, S_EXP)

// one dependent chain (latency)
__global__ void k_fma_dep(float *out, Stamp *st, int iters, float a, float b)
{
    float acc = (float)threadIdx.x * 1e-3f;
    Stamp s;
    stamp_begin(s);
    for (int it = 0; it < iters; ++it) {
#define S(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        REP8(S)
#undef S
    }
    stamp_end(s, st);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// the backward's pixel: 7 VALU + 1 SALU, one row of 8 pixels is a serial chain in G / rt and in the three moments
__global__ void k_bwd_row(float *out, Stamp *st, int iters, float a, float b)
{
    float G = a, rt = b, rr = 0.99999f, g[8], t0 = 0.f, t1 = 0.f, t2 = 0.f, w;
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = (float)threadIdx.x * 1e-3f + i;
    const float thr = 1e-30f;
    unsigned long long ex;
    asm volatile("s_mov_b64 %0, exec" : "=s"(ex));
    Stamp s;
    stamp_begin(s);
    for (int it = 0; it < iters; ++it) {
#define S(c)                                                                                                              \
        asm volatile("v_cmpx_le_f32_e32 %[thr], %[G]\n\t"                                                                 \
                     "v_mul_f32_e32 %[w], %[G], %[g]\n\t"                                                                 \
                     "v_add_f32_e32 %[t0], %[t0], %[w]\n\t"                                                               \
                     "v_fmac_f32_e32 %[t1], 0x40200000, %[w]\n\t"                                                         \
                     "v_fmac_f32_e32 %[t2], 0x40c80000, %[w]\n\t"                                                         \
                     "s_mov_b64 exec, %[ex]\n\t"                                                                          \
                     "v_mul_f32_e32 %[G], %[G], %[rt]\n\t"                                                                \
                     "v_mul_f32_e32 %[rt], %[rt], %[rr]"                                                                  \
                     : [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [w] "=&v"(w), [G] "+v"(G), [rt] "+v"(rt)              \
                     : [thr] "v"(thr), [g] "v"(g[c]), [rr] "v"(rr), [ex] "s"(ex)                                          \
                     : "vcc");
        REP8(S)
#undef S
    }
    stamp_end(s, st);
    out[blockIdx.x * blockDim.x + threadIdx.x] = t0 + t1 + t2 + G + rt;
}
// two independent rows interleaved (what a 2-row software schedule would look like to the issue logic)
__global__ void k_bwd_row2(float *out, Stamp *st, int iters, float a, float b)
{
    float G = a, rt = b, H = a * 0.5f, ht = b, rr = 0.99999f, g[8], t0 = 0.f, t1 = 0.f, t2 = 0.f, u0 = 0.f, u1 = 0.f, u2 = 0.f, w, x;
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = (float)threadIdx.x * 1e-3f + i;
    const float thr = 1e-30f;
    unsigned long long ex;
    asm volatile("s_mov_b64 %0, exec" : "=s"(ex));
    Stamp s;
    stamp_begin(s);
    for (int it = 0; it < iters; ++it) {
#define S(c)                                                                                                              \
        asm volatile("v_cmpx_le_f32_e32 %[thr], %[G]\n\t"                                                                 \
                     "v_mul_f32_e32 %[w], %[G], %[g]\n\t"                                                                 \
                     "v_add_f32_e32 %[t0], %[t0], %[w]\n\t"                                                               \
                     "v_fmac_f32_e32 %[t1], 0x40200000, %[w]\n\t"                                                         \
                     "v_fmac_f32_e32 %[t2], 0x40c80000, %[w]\n\t"                                                         \
                     "s_mov_b64 exec, %[ex]\n\t"                                                                          \
                     "v_mul_f32_e32 %[G], %[G], %[rt]\n\t"                                                                \
                     "v_mul_f32_e32 %[rt], %[rt], %[rr]\n\t"                                                              \
                     "v_cmpx_le_f32_e32 %[thr], %[H]\n\t"                                                                 \
                     "v_mul_f32_e32 %[x], %[H], %[g]\n\t"                                                                 \
                     "v_add_f32_e32 %[u0], %[u0], %[x]\n\t"                                                               \
                     "v_fmac_f32_e32 %[u1], 0x40200000, %[x]\n\t"                                                         \
                     "v_fmac_f32_e32 %[u2], 0x40c80000, %[x]\n\t"                                                         \
                     "s_mov_b64 exec, %[ex]\n\t"                                                                          \
                     "v_mul_f32_e32 %[H], %[H], %[ht]\n\t"                                                                \
                     "v_mul_f32_e32 %[ht], %[ht], %[rr]"                                                                  \
                     : [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [w] "=&v"(w), [G] "+v"(G), [rt] "+v"(rt), [u0] "+v"(u0),  \
                       [u1] "+v"(u1), [u2] "+v"(u2), [x] "=&v"(x), [H] "+v"(H), [ht] "+v"(ht)                              \
                     : [thr] "v"(thr), [g] "v"(g[c]), [rr] "v"(rr), [ex] "s"(ex)                                          \
                     : "vcc");
        S(0) S(1) S(2) S(3)
#undef S
    }
    stamp_end(s, st);
    out[blockIdx.x * blockDim.x + threadIdx.x] = t0 + t1 + t2 + G + rt + u0 + u1 + u2 + H + ht;
}
// the backward's pixel with compare + select instead of the EXEC mask (8 VALU, no SALU)
__global__ void k_bwd_row_sel(float *out, Stamp *st, int iters, float a, float b)
{
    float G = a, rt = b, rr = 0.99999f, g[8], t0 = 0.f, t1 = 0.f, t2 = 0.f, w;
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = (float)threadIdx.x * 1e-3f + i;
    const float thr = 1e-30f;
    Stamp s;
    stamp_begin(s);
    for (int it = 0; it < iters; ++it) {
#define S(c)                                                                                                              \
        asm volatile("v_mul_f32_e32 %[w], %[G], %[g]\n\t"                                                                 \
                     "v_cmp_le_f32_e32 vcc, %[thr], %[G]\n\t"                                                             \
                     "v_mul_f32_e32 %[G], %[G], %[rt]\n\t"                                                                \
                     "v_cndmask_b32_e32 %[w], 0, %[w], vcc\n\t"                                                           \
                     "v_mul_f32_e32 %[rt], %[rt], %[rr]\n\t"                                                              \
                     "v_add_f32_e32 %[t0], %[t0], %[w]\n\t"                                                               \
                     "v_fmac_f32_e32 %[t1], 0x40200000, %[w]\n\t"                                                         \
                     "v_fmac_f32_e32 %[t2], 0x40c80000, %[w]"                                                             \
                     : [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [w] "=&v"(w), [G] "+v"(G), [rt] "+v"(rt)              \
                     : [thr] "v"(thr), [g] "v"(g[c]), [rr] "v"(rr)                                                        \
                     : "vcc");
        REP8(S)
#undef S
    }
    stamp_end(s, st);
    out[blockIdx.x * blockDim.x + threadIdx.x] = t0 + t1 + t2 + G + rt;
}
// the forward's pixel: cmp + cndmask + add + 2 recurrence muls, 8 accumulators per row
__global__ void k_fwd_row(float *out, Stamp *st, int iters, float a, float b)
{
    float G = a, rt = b, rr = 0.99999f, acc[8], w;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    const float thr = 1e-30f;
    Stamp s;
    stamp_begin(s);
    for (int it = 0; it < iters; ++it) {
#define S(c)                                                                                                              \
        asm volatile("v_cmp_le_f32_e32 vcc, %[thr], %[G]\n\t"                                                             \
                     "v_cndmask_b32_e32 %[w], 0, %[G], vcc\n\t"                                                           \
                     "v_mul_f32_e32 %[G], %[G], %[rt]\n\t"                                                                \
                     "v_add_f32_e32 %[acc], %[acc], %[w]\n\t"                                                             \
                     "v_mul_f32_e32 %[rt], %[rt], %[rr]"                                                                  \
                     : [acc] "+v"(acc[c]), [w] "=&v"(w), [G] "+v"(G), [rt] "+v"(rt)                                       \
                     : [thr] "v"(thr), [rr] "v"(rr)                                                                       \
                     : "vcc");
        REP8(S)
#undef S
    }
    stamp_end(s, st);
    float r = G + rt;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
// backward row + the two ds_read_b128 of dL/dpix it is fed by
__global__ void k_bwd_row_lds(float *out, Stamp *st, int iters, float a, float b)
{
    __shared__ float gt[4 * 324];
    for (int i = threadIdx.x; i < 4 * 324; i += blockDim.x) gt[i] = (float)i * 1e-3f;
    __syncthreads();
    float G = a, rt = b, rr = 0.99999f, t0 = 0.f, t1 = 0.f, t2 = 0.f, w;
    const float thr = 1e-30f;
    unsigned long long ex;
    asm volatile("s_mov_b64 %0, exec" : "=s"(ex));
    const float *base = gt + (threadIdx.x & 3) * 8 + ((threadIdx.x >> 2) & 1) * 160 + ((threadIdx.x >> 3) % 3) * 324;
    Stamp s;
    stamp_begin(s);
    for (int it = 0; it < iters; ++it) {
        const float4 v0 = *reinterpret_cast<const float4 *>(base + (it & 7) * 20);
        const float4 v1 = *reinterpret_cast<const float4 *>(base + (it & 7) * 20 + 4);
        const float g[8] = { v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w };
#define S(c)                                                                                                              \
        asm volatile("v_cmpx_le_f32_e32 %[thr], %[G]\n\t"                                                                 \
                     "v_mul_f32_e32 %[w], %[G], %[g]\n\t"                                                                 \
                     "v_add_f32_e32 %[t0], %[t0], %[w]\n\t"                                                               \
                     "v_fmac_f32_e32 %[t1], 0x40200000, %[w]\n\t"                                                         \
                     "v_fmac_f32_e32 %[t2], 0x40c80000, %[w]\n\t"                                                         \
                     "s_mov_b64 exec, %[ex]\n\t"                                                                          \
                     "v_mul_f32_e32 %[G], %[G], %[rt]\n\t"                                                                \
                     "v_mul_f32_e32 %[rt], %[rt], %[rr]"                                                                  \
                     : [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [w] "=&v"(w), [G] "+v"(G), [rt] "+v"(rt)              \
                     : [thr] "v"(thr), [g] "v"(g[c]), [rr] "v"(rr), [ex] "s"(ex)                                          \
                     : "vcc");
        REP8(S)
#undef S
    }
    stamp_end(s, st);
    out[blockIdx.x * blockDim.x + threadIdx.x] = t0 + t1 + t2 + G + rt;
}

template <typename K>
static void run(const char *name, K kern, int wps, int block, int iters, double valu_per_iter, float *out, Stamp *dst)
{
    // wps waves per SIMD wanted: per CU 4*wps waves = 4*wps*64/block workgroups
    const int wg_per_cu = 4 * wps * 64 / block;
    const int grid = 256 * wg_per_cu, nw = grid * block / 64;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    kern<<<grid, block>>>(out, dst, iters / 8, 1.00001f, 0.99999f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    kern<<<grid, block>>>(out, dst, iters, 1.00001f, 0.99999f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamp> st(nw);
    CHECK(hipMemcpy(st.data(), dst, sizeof(Stamp) * nw, hipMemcpyDeviceToHost));
    // waves per SIMD as dispatched
    std::map<unsigned, int> per_simd;
    for (auto &s : st) per_simd[((s.xcc & 0xf) << 16) | (s.hw & 0xfff0)]++;   // xcc | se, sh, cu, pipe, simd (wave slot masked out)
    int wmin = 1 << 30, wmax = 0;
    for (auto &kv : per_simd) { wmin = std::min(wmin, kv.second); wmax = std::max(wmax, kv.second); }
    std::vector<double> cpi, ghz;
    unsigned long long rmin = ~0ull, rmax = 0;
    for (auto &s : st) {
        const double dc = (double)(s.c1 - s.c0), dr = (double)(s.r1 - s.r0);
        const int w_here = per_simd[((s.xcc & 0xf) << 16) | (s.hw & 0xfff0)];
        cpi.push_back(dc / ((double)iters * valu_per_iter * w_here));
        ghz.push_back(dc / dr * 0.1);
        rmin = std::min(rmin, s.r0);
        rmax = std::max(rmax, s.r1);
    }
    std::sort(cpi.begin(), cpi.end());
    std::sort(ghz.begin(), ghz.end());
    printf("%-16s want %d w/SIMD  got %d..%d on %4zu SIMDs | %5.2f cyc/VALU/SIMD (med; p10 %5.2f p90 %5.2f) | clock %5.3f GHz (p10 %5.3f p90 %5.3f) | "
           "kernel %7.3f ms by stamps, %7.3f ms by events\n",
           name, wps, wmin, wmax, per_simd.size(), cpi[cpi.size() / 2], cpi[cpi.size() / 10], cpi[cpi.size() * 9 / 10], ghz[ghz.size() / 2],
           ghz[ghz.size() / 10], ghz[ghz.size() * 9 / 10], (double)(rmax - rmin) * 1e-5, ms);
}

int main()
{
    float *out;
    Stamp *dst;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * 64 * 256));
    CHECK(hipMalloc(&dst, sizeof(Stamp) * 256 * 64 * 4));
    const int IT = 1 << 15;   // x 8 instructions x 2+ cycles x waves >= 1 ms at 4 waves/SIMD
    for (int w : { 1, 2, 4, 8 }) {
        run("v_fma_f32", k_fma, w, 256, IT, 8, out, dst);
        run("v_fma_f32 dep", k_fma_dep, w, 256, IT, 8, out, dst);
        run("v_mul_f32", k_mul, w, 256, IT, 8, out, dst);
        run("v_exp_f32", k_exp, w, 256, IT, 8, out, dst);
        run("bwd row cmpx", k_bwd_row, w, 64, IT / 4, 56, out, dst);
        run("bwd 2 rows", k_bwd_row2, w, 64, IT / 4, 56, out, dst);
        run("bwd row select", k_bwd_row_sel, w, 64, IT / 4, 64, out, dst);
        run("bwd row + lds", k_bwd_row_lds, w, 64, IT / 4, 56, out, dst);
        run("fwd row", k_fwd_row, w, 256, IT / 4, 40, out, dst);
    }
    return 0;
}
