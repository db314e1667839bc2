// raster_geom.hip -- per-Gaussian kernels of the X-ray rasterizer (HBM-bound streams, one lane per
// Gaussian): projection + ray-space covariance + tile rectangle (forward), key/value emission, and the
// fused geometry backward.  Compiled with -ffp-contract=off (see r2_math.hpp).
//
// Reference: RAS/forward.cu:77-289, RAS/auxiliary.h:45-60,143-168, RAS/rasterizer_impl.cu:54-111,
//            RAS/backward.cu:145-444.
#include "r2_math.hpp"
#include "raster_state.hpp"

R2_TS_DEFINE(geom)

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace r2 {

// pixel coordinate of an NDC coordinate, evaluated in double like the reference (RAS/auxiliary.h:45-48)
__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

struct Cov2D {
    float tx, ty, tz;
    float xmul, ymul;
    M3 J, W, M, cov;
};

// t (clamped), J, W, M = W*J, cov = M^T Vrk^T M -- forward RAS/forward.cu:77-131, backward RAS/backward.cu:165-219
__device__ __forceinline__ void cov2d_common(float3 mean, float fx, float fy, float tan_fovx, float tan_fovy,
                                             const float *cov3D, const float *__restrict__ view, int mode, Cov2D &c)
{
    float3 t = xform4x3(mean, view);
    if (mode == 0) {
        const float limx = 1.3f, limy = 1.3f;
        t.x = fminf(limx, fmaxf(-limx, t.x));
        t.y = fminf(limx, fmaxf(-limx, t.y));
        c.xmul = (t.x < -limx || t.x > limx) ? 0.f : 1.f;
        c.ymul = (t.y < -limy || t.y > limy) ? 0.f : 1.f;
        c.J = m3(fx, 0.0f, 0.0f, 0.0f, fy, 0.0f, 0.0f, 0.0f, 1.0f);
    } else {
        const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
        const float txtz = t.x / t.z, tytz = t.y / t.z;
        t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
        t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
        c.xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
        c.ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
        const float l = sqrtf(t.x * t.x + t.y * t.y + t.z * t.z);
        c.J = m3(fx / t.z, 0.0f, -(fx * t.x) / (t.z * t.z), 0.0f, fy / t.z, -(fy * t.y) / (t.z * t.z), t.x / l, t.y / l,
                 t.z / l);
    }
    c.tx = t.x; c.ty = t.y; c.tz = t.z;
    c.W = m3(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    c.M = mul(c.W, c.J);
    const M3 Vrk = m3(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    c.cov = mul(mul(tr(c.M), tr(Vrk)), c.M);
}

__device__ __forceinline__ float mu_of(float circ, float diamond)
{
    // 2*M_PI is a double in the reference: the quotient is formed in double, narrowed afterwards
    const double q = 2 * M_PI * (double)circ / (double)diamond;
    float mu = 0.0f;
    if ((float)q > 0.0f) mu = (float)sqrt(q);
    return mu;
}

// ------------------------------------------------------------------ forward: one lane per Gaussian
// returns the Gaussian's depth key (DEPTH_CULLED_KEY if it emits nothing) and its number of tiles
__device__ __forceinline__ void raster_preprocess_one(
    int idx /* view instance v * P + src */, int src /* Gaussian */, const float *__restrict__ means3D, const float *__restrict__ scales, float scale_modifier,
    const float *__restrict__ rotations, const float *__restrict__ opacities, const float *__restrict__ cov3D_precomp,
    const float *__restrict__ view, const float *__restrict__ proj, int W, int H, float tan_fovx, float tan_fovy,
    float focal_x, float focal_y, int mode, int gx, int gy,
    int *__restrict__ radii, float4 *__restrict__ rec, uint32_t *__restrict__ depth_key,
    float *__restrict__ cov3Ds, uint32_t *__restrict__ tiles_touched, float2 *__restrict__ op_mu,
    uint32_t *__restrict__ thin_flag, const DepthReg &reg, uint32_t &key_out, uint2 &bt_out, uint32_t &rect_out)
{
    key_out = DEPTH_CULLED_KEY;
    radii[idx] = 0;
    tiles_touched[idx] = 0;
    depth_key[idx] = 0xFFFFFFFFu;   // culled Gaussians sort behind every visible one

    const float3 p = make_float3(means3D[3 * src], means3D[3 * src + 1], means3D[3 * src + 2]);
    const float3 p_view = xform4x3(p, view);
    if (p_view.z <= 0.2f) return;   // near cull (RAS/auxiliary.h:158)
    const float4 p_hom = xform4x4(p, proj);
    const float p_w = 1.0f / (p_hom.w + 0.0000001f);
    const float projx = p_hom.x * p_w, projy = p_hom.y * p_w;

    float cov3D[6];
    if (cov3D_precomp != nullptr) {
#pragma unroll
        for (int k = 0; k < 6; ++k) cov3D[k] = cov3D_precomp[6 * src + k];
    } else {
        const float4 q = reinterpret_cast<const float4 *>(rotations)[src];
        cov3d_from_scale_rot(scales[3 * src], scales[3 * src + 1], scales[3 * src + 2], scale_modifier, q, cov3D);
        // stored for the backward like the reference's geometry state (written even if rejected below, Q12)
        if (cov3Ds != nullptr)
#pragma unroll
            for (int k = 0; k < 6; ++k) cov3Ds[6 * idx + k] = cov3D[k];
    }

    Cov2D c;
    cov2d_common(p, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, view, mode, c);
    const float a = c.cov.m[0][0], b = c.cov.m[0][1], cc = c.cov.m[0][2];
    const float d = c.cov.m[1][1], e = c.cov.m[1][2], f = c.cov.m[2][2];
    const float diamond = a * d - b * b;
    const float circ = a * d * f + 2 * b * cc * e - a * e * e - f * b * b - d * cc * cc;
    const float mu = mu_of(circ, diamond);

    const float det = (a * d - b * b);
    if (det == 0.0f) return;
    const float det_inv = 1.f / det;
    const float conA = d * det_inv, conB = -b * det_inv, conC = a * det_inv;

    const float mid = 0.5f * (a + d);
    const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
    const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
    const float px = ndc2pix(projx, W), py = ndc2pix(projy, H);
    int x0, y0, x1, y1;
    tile_rect(px, py, (int)my_radius, gx, gy, x0, y0, x1, y1);
    if ((uint32_t)(x1 - x0) * (uint32_t)(y1 - y0) == 0) return;

    depth_key[idx] = __float_as_uint(p_view.z);   // z > 0.2: float order == unsigned order of the bits
    radii[idx] = (int)my_radius;
    tiles_touched[idx] = (uint32_t)(y1 - y0) * (uint32_t)(x1 - x0);
    key_out = __float_as_uint(p_view.z);
    // hinted depth order: the key goes straight into its bucket; the ticket comes back while the record is computed
    bt_out = depth_register_key(reg, key_out, (uint32_t)(y1 - y0) * (uint32_t)(x1 - x0));
    rect_out = depth_rect_pack(x0, y0, x1 - x0, y1 - y0);   // (meaningful for grids up to 256 x 256 tiles: the caller checks)
    // 32-byte render record: centre, conic pre-scaled so that the render kernels evaluate
    // alpha = opacity*mu*exp(power) as exp2(A2 dx^2 + B2 dx dy + C2 dy^2 + L), L = log2(opacity*mu), and the
    // half-extents (hx, hy) of the bounding box of the set where alpha can reach the reference's 1e-5 cut-off
    // (RAS/forward.cu:374) -- the render kernels skip 8x8 pixel blocks that lie outside it.
    const float op = opacities[src];
    const float opmu = op * mu;
    const float L = opmu > 0.0f ? log2f(opmu) : -INFINITY;
    float hx = INFINITY, hy = INFINITY;   // +inf: never cull (degenerate / ill-conditioned conics)
    {
        // alpha >= 1e-5  <=>  q = A dx^2 + 2 B dx dy + C dy^2 <= 2 ln2 (L - log2(1e-5)); the bounding box of that
        // ellipse has half-widths sqrt(qmax * C / det), sqrt(qmax * A / det), computed in double from the float
        // conic the kernels actually evaluate, then padded (0.4 % + 0.05 px) against float rounding of q.
        const double qmax = 2.0 * (double)LN2 * ((double)L - (double)LOG2_ALPHA_MIN_2D) + 1e-3;
        const double A = conA, B = conB, C = conC;
        const double det2 = A * C - B * B, trc = A + C;
        if (!(qmax > 0.0)) {
            hx = hy = -INFINITY;   // opacity*mu below the cut-off: no pixel can pass (exp(power) <= 1)
        } else if (det2 > 0.0 && A > 0.0 && C > 0.0 && trc * trc <= 1.0e4 * det2) {
            const double ex = sqrt(qmax * C / det2) * 1.004 + 0.05, ey = sqrt(qmax * A / det2) * 1.004 + 0.05;
            if (ex < 1.0e30 && ey < 1.0e30) { hx = (float)ex; hy = (float)ey; }
        }
    }
    rec[2 * idx] = make_float4(px, py, (-0.5f * LOG2E) * conA, (-LOG2E) * conB);
    rec[2 * idx + 1] = make_float4((-0.5f * LOG2E) * conC, L, hx, hy);
    op_mu[idx] = make_float2(op, mu);
    // counted for the host (DW_USER): the visible Gaussians the render kernels evaluate exactly, pixel by pixel, because the
    // recurrences are not safe for them (item_tier; rounds 1-5: the flag that selected the re-anchoring kernel variant)
    if (thin_flag && hx > 0.f && item_tier((-0.5f * LOG2E) * conA, (-LOG2E) * conB, (-0.5f * LOG2E) * conC, L, hx, hy) != 0) *thin_flag = 1u;
}

__global__ void __launch_bounds__(256) raster_preprocess_kernel(
    int P, int V, const float *__restrict__ means3D, const float *__restrict__ scales, float scale_modifier,
    const float *__restrict__ rotations, const float *__restrict__ opacities, const float *__restrict__ cov3D_precomp,
    const float *__restrict__ views, const float *__restrict__ projs, int W, int H, float tan_fovx, float tan_fovy,
    float focal_x, float focal_y, int mode, int gx, int gy,
    int *__restrict__ radii, float4 *__restrict__ rec, uint32_t *__restrict__ depth_key,
    float *__restrict__ cov3Ds, uint32_t *__restrict__ tiles_touched, float2 *__restrict__ op_mu,
    uint32_t *__restrict__ thin_flag, DepthReg reg)
{
    // one lane per VIEW INSTANCE: idx = v * P + src (V = 1: the reference's one lane per Gaussian); view v's matrices
    const int idx = blockIdx.x * 256 + threadIdx.x;
    R2_TS_AT(geom, 0);
    uint32_t key = DEPTH_CULLED_KEY;
    uint2 bt = make_uint2(0u, 0u);
    uint32_t rect = 0u, thin = 0u;
    if (idx < P * V) {
        const int v = V == 1 ? 0 : idx / P;
        raster_preprocess_one(idx, idx - v * P, means3D, scales, scale_modifier, rotations, opacities, cov3D_precomp, views + 16 * v,
                              projs + 16 * v, W, H, tan_fovx, tan_fovy, focal_x, focal_y, mode, gx, gy, radii, rec, depth_key, cov3Ds,
                              tiles_touched, op_mu, &thin, reg, key, bt, rect);
    }
    // how many Gaussians need the re-anchored row recurrence: a COUNT (one atomic per wave that holds any), like the tile-first chain's
    if (thin_flag != nullptr) {
        const uint32_t nthin = (uint32_t)__popcll(__ballot(thin != 0u));
        if (nthin != 0u && (threadIdx.x & 63) == 0) atomicAdd(thin_flag, nthin);
    }
    R2_TS_AT(geom, 1);
    depth_register_end(reg, (uint32_t)idx, key, bt, rect);
    R2_TS_AT(geom, 2);
}

// ---- tile-first binning (rounds 4-5; raster_tilefirst.hip has the rest of the chain and the rationale).  The same per-Gaussian
// work as above, without the depth-bucket registration; instead the workgroup
//   * counts its instances per tile in an LDS histogram (one LDS atomic per instance) and adds every non-zero count to the
//     global per-tile counter with ONE returning atomic per (workgroup, tile) it touches -- the returned value is where this
//     workgroup's instances start inside that tile's list, kept in tf_wgoff[workgroup][tile] for the scatter kernel;
//   * hands every visible Gaussian its run of backward scratch rows: a workgroup scan of tiles_touched + one returning 64-bit
//     atomic {visible << 40 | instances} for the workgroup's base (any disjoint assignment serves the backward);
//   * leaves its key range in its slot of tf_wgmm.
// Round 5: (1) no epilogue.  Round 4's last workgroup to finish (a `done` counter behind s_waitcnt vmcnt(0)) reduced the key
// ranges and posted the totals to the host: 4.1 us between the end of the slowest workgroup and the end of the kernel, on the
// critical path of everything (stamps: profiles/experiments/r05_round4_chain_stamps.txt).  The totals now simply stay in the
// counters and workgroup 0 of the NEXT kernel -- which starts when this one has drained anyway -- posts them.  (2) the Gaussians
// are dealt evenly to one workgroup per CU (tf_grid): a thread owns up to TF_PER_THREAD_MAX of them.
// (Measured and left out, round 5, same box, alternating runs against this kernel at 22.8 us of stage time: requesting the inputs of
// BOTH of a thread's Gaussians before computing the first one: the arithmetic phase 5.3 -> 6.8 us by the stamps -- eleven more live
// registers through 150 lines of double-precision culling arithmetic -- kernel 25.0 us; walking rectangles of more than 8 tiles with
// the whole wave, 64 tiles per step, here and in the scatter kernel: histogram phase 3.9 -> 4.6 us, scatter 13.5 -> 14.2 us: a wave
// holds only a handful of such rectangles and each costs three v_readlane + a division-free index computation per step.)
// SHARE: more producer workgroups than CUs (clouds beyond 524k Gaussians): compiled for <= 64 VGPRs (four spilled), so that two
// 1024-thread workgroups run side by side on a CU instead of one after the other; the plain instantiation (68 VGPRs) has its CU to itself.
// MV: stacked views.  A single view is compiled without the view index and its division (round 6: with them in the SHARE
// instantiation, the 64-register cap spilled 35 registers instead of 4 and the preprocess of 1M Gaussians took 80 us instead of 52).
template <bool SL /* depth slabs in use (TFSlabs): the plain instantiation carries none of their arithmetic */, bool SHARE, bool MV>
__global__ void __launch_bounds__(TF_THREADS_MAX, SHARE ? 8 : 4) raster_preprocess_tf_kernel(
    int P /* per view */, int V /* stacked views (round 6): the kernel runs over the V * P view instances v * P + i, tile grids stacked */,
    uint32_t per_wg, const TFSlabs slabs, const float *__restrict__ means3D, const float *__restrict__ scales, float scale_modifier,
    const float *__restrict__ rotations, const float *__restrict__ opacities, const float *__restrict__ cov3D_precomp,
    const float *__restrict__ view, const float *__restrict__ proj, int W, int H, float tan_fovx, float tan_fovy,
    float focal_x, float focal_y, int mode, int gx, int gy,
    int *__restrict__ radii, float4 *__restrict__ rec, uint32_t *__restrict__ depth_key,
    float *__restrict__ cov3Ds, uint32_t *__restrict__ tiles_touched, float2 *__restrict__ op_mu,
    uint32_t *__restrict__ first, uint32_t *__restrict__ rects, uint32_t *__restrict__ wgoff, uint32_t *__restrict__ wgmm,
    TFCounters *__restrict__ ctr)
{
    extern __shared__ uint32_t tf_hist[];   // [lists = T x slabs]
    constexpr int MAXW = (int)(TF_THREADS_MAX / 64), NI = (int)TF_PER_THREAD_MAX;
    __shared__ uint32_t s_wn[NI][MAXW], s_wv[MAXW], s_kmx[MAXW], s_nkmn[MAXW], s_thin, s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t NT = blockDim.x;
    const int nw = (int)(NT >> 6);
    const uint32_t nsl = SL ? slabs.n : 1u;
    const uint32_t T = (uint32_t)(gx * gy * (MV ? V : 1)) * nsl;   // lists
    const uint32_t PV = (uint32_t)P * (uint32_t)(MV ? V : 1);
    const uint32_t g0 = blockIdx.x * per_wg, g1 = min(g0 + per_wg, PV);   // this workgroup's view instances
    R2_TS_AT(geom, 0);
    for (uint32_t t = tid; t < T; t += NT) tf_hist[t] = 0u;
    if (tid == 0) s_thin = 0u;
    uint32_t key[NI], rect[NI], thin = 0u;
    const DepthReg noreg{};
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        key[it] = DEPTH_CULLED_KEY; rect[it] = 0u;
        const uint32_t idx = g0 + (uint32_t)it * NT + (uint32_t)tid;
        uint2 bt = make_uint2(0u, 0u);
        if ((it == 0 || NT * (uint32_t)it < per_wg) && idx < g1) {   // (workgroup-uniform first half: no second round for small workgroups)
            const uint32_t v = MV ? idx / (uint32_t)P : 0u;   // the instance's view: its matrices, its rows of the stacked grid
            raster_preprocess_one((int)idx, (int)(idx - v * (uint32_t)P), means3D, scales, scale_modifier, rotations, opacities,
                                  cov3D_precomp, view + 16u * v, proj + 16u * v, W, H, tan_fovx, tan_fovy, focal_x, focal_y, mode, gx, gy,
                                  radii, rec, depth_key, cov3Ds, tiles_touched, op_mu, &thin, noreg, key[it], bt, rect[it]);
            if (MV && key[it] != DEPTH_CULLED_KEY) rect[it] += (v * (uint32_t)gy) << 8;   // y0 of the rectangle in the stacked grid (< 256 rows)
        }
    }
    R2_TS_AT(geom, 10);
    __syncthreads();                           // the histogram is clear
    uint32_t n[NI], incl[NI], nvis = 0u, kmx = 0u, nkmn = 0u;
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const bool vis = key[it] != DEPTH_CULLED_KEY;
        n[it] = vis ? depth_rect_count(rect[it]) : 0u;
        if (vis) {
            const uint32_t x0 = rect[it] & 0xFFu, y0 = (rect[it] >> 8) & 0xFFu, w = ((rect[it] >> 16) & 0xFFu) + 1u, h = (rect[it] >> 24) + 1u;
            const uint32_t sl = SL ? tf_slab_of(key[it], slabs) : 0u;
            for (uint32_t r = 0; r < h; ++r) {
                const uint32_t row = ((y0 + r) * (uint32_t)gx + x0) * nsl + sl;
                for (uint32_t c = 0; c < w; ++c) atomicAdd(&tf_hist[row + c * nsl], 1u);
            }
            rects[g0 + (uint32_t)it * NT + (uint32_t)tid] = rect[it];
            kmx = max(kmx, key[it]);
            nkmn = max(nkmn, ~key[it]);
        }
        // exclusive prefix of n inside the workgroup, in (round, thread) order
        incl[it] = n[it];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl[it], d);
            if (lane >= d) incl[it] += up;
        }
        nvis += (uint32_t)__popcll(__ballot(vis));
        if (lane == 63) s_wn[it][wave] = incl[it];
    }
    // key range of the call (the sort kernel's depth histograms are laid over it)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        kmx = max(kmx, (uint32_t)__shfl_xor(kmx, d));
        nkmn = max(nkmn, (uint32_t)__shfl_xor(nkmn, d));
    }
    if (lane == 63) { s_wv[wave] = nvis; s_kmx[wave] = kmx; s_nkmn[wave] = nkmn; }
    // how many Gaussians need the re-anchored row recurrence (round 6: a COUNT -- the forward's plain variant serves a few of them
    // through its exact path; only a scene with many takes the re-anchoring variant, see raster_forward_tilefirst)
    {
        const uint32_t nthin = (uint32_t)__popcll(__ballot(thin != 0u));
        if (nthin != 0u && lane == 0) atomicAdd(&s_thin, nthin);
    }
    R2_TS_AT(geom, 11);
    __syncthreads();                           // the histogram is complete; wave sums are in place
    R2_TS_AT(geom, 12);
    uint32_t before[NI], tn = 0, tv = 0;
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        before[it] = tn;                       // everything of the earlier rounds ...
        for (int q = 0; q < nw; ++q) {
            if (q < wave) before[it] += s_wn[it][q];   // ... and the earlier waves of this one
            tn += s_wn[it][q];
        }
    }
    for (int q = 0; q < nw; ++q) tv += s_wv[q];
    unsigned long long base_old = 0ull;
    if (tid == 0) {
        // (its result is consumed at the very end: the round trip of this same-address atomic -- every workgroup bumps it --
        // runs behind the per-tile atomics below)
        base_old = atomicAdd(&ctr->total, ((unsigned long long)tv << 40) | (unsigned long long)tn);
        if (s_thin) atomicAdd(&ctr->thin, s_thin);
        // the workgroup's key range: one slot per workgroup, reduced by the next kernel (atomicMax on ONE word from every
        // workgroup was measured: +5 us on the kernel, same-address atomics retire at ~90 per microsecond)
        uint32_t a = 0u, b = 0u;
        for (int q = 0; q < nw; ++q) { a = max(a, s_kmx[q]); b = max(b, s_nkmn[q]); }
        wgmm[2u * blockIdx.x] = a;
        wgmm[2u * blockIdx.x + 1u] = b;
    }
    // one returning atomic per tile this workgroup touches: where its instances start inside the tile's list
    uint32_t *__restrict__ my_off = wgoff + (size_t)blockIdx.x * T;
    for (uint32_t t = tid; t < T; t += NT) {
        const uint32_t c = tf_hist[t];
        if (c) my_off[t] = atomicAdd(&ctr->tile_count[t], c);
    }
    R2_TS_AT(geom, 13);
    if (tid == 0) s_base = (uint32_t)base_old;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NI; ++it)
        if (key[it] != DEPTH_CULLED_KEY) first[g0 + (uint32_t)it * NT + (uint32_t)tid] = s_base + before[it] + incl[it] - n[it];
    R2_TS_AT(geom, 1);
}

// z_view > 0.2 mask (RAS/rasterizer_impl.cu:54-66)
__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float *__restrict__ means3D,
                                                           const float *__restrict__ view, uint8_t *__restrict__ present)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float3 p = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
    present[idx] = xform4x3(p, view).z <= 0.2f ? 0 : 1;
}

// Instance emission (the reference's duplicateWithKeys, RAS/rasterizer_impl.cu:70-111), in DEPTH order:
// sorted position j -> Gaussian order[j] -> its tiles, y-major / x-minor, written at offsets[j-1]...
// Only the tile id is emitted as the sort key: the list is already depth-ordered, so a stable sort by
// tile reproduces the reference's (tile | depth) order exactly (see binning.hip).
// One WAVE serves 64 consecutive sorted positions: their output runs are contiguous, so the wave walks that
// span 64 instances at a time (coalesced stores) and each lane finds its owner with a 6-step search over
// the lanes' exclusive offsets (ds_bpermute), instead of every lane dribbling out its own run.
__global__ void __launch_bounds__(256) raster_duplicate_kernel(
    int P /* view instances */, int Pview, const float4 *__restrict__ rec, const uint32_t *__restrict__ order, const uint32_t *__restrict__ offsets,
    const int *__restrict__ radii, int gx, int gy, uint32_t *__restrict__ first, uint32_t *__restrict__ tiles,
    uint32_t *__restrict__ vals, const uint32_t *__restrict__ nvis)
{
    R2_TS_AT(geom, 4);
    // several views in this call?  Decided on the caller's counts, BEFORE P is trimmed to the visible prefix: a batched call
    // whose visible count happened to equal the per-view Gaussian count would otherwise be taken for a single view and every
    // instance would get view 0's tile rows
    const bool multi_view = Pview != P;
    if (nvis) P = min(P, (int)*nvis);   // hinted depth order: only the visible prefix of order / offsets is written
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave_first = j - lane;
    if (wave_first >= P) return;
    uint32_t id = 0;
    int rad = 0;
    if (j < P) {
        id = order[j];
        rad = radii[id];
    }
    const bool live = rad > 0;
    // exclusive offset of this lane's run; dead lanes get the running offset so the search stays monotone
    uint32_t excl = 0, incl = 0;
    if (j < P) {
        incl = offsets[j];
        excl = j == 0 ? 0u : offsets[j - 1];
    } else {
        incl = excl = offsets[P - 1];
    }
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    if (live) {
        const float4 r0 = rec[2 * id];
        tile_rect(r0.x, r0.y, rad, gx, gy, x0, y0, x1, y1);
        first[id] = excl;   // where this Gaussian's instance run starts: the backward's scratch rows
    }
    const uint32_t wbeg = __shfl(excl, 0);
    const int last_lane = min(63, P - 1 - wave_first);
    const uint32_t wend = __shfl(incl, last_lane);
    const int rw = x1 - x0;
    for (uint32_t base = wbeg; base < wend; base += 64) {
        const uint32_t k = base + lane;
        // owner = last lane whose exclusive offset is <= k  (binary search over 64 lanes)
        int lo = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {
            const int probe = lo + step;
            const uint32_t e = __shfl(excl, probe & 63);
            if (probe <= last_lane && e <= k) lo = probe;
        }
        const uint32_t o_excl = __shfl(excl, lo);
        const int o_x0 = __shfl(x0, lo), o_y0 = __shfl(y0, lo), o_rw = __shfl(rw, lo);
        const uint32_t o_id = __shfl(id, lo);
        if (k < wend) {
            const uint32_t local = k - o_excl;
            const int ty = o_y0 + (int)(local / (uint32_t)o_rw);
            const int tx = o_x0 + (int)(local % (uint32_t)o_rw);
            // the views' tile grids are stacked: view v = o_id / Pview owns tile rows [v * gy, (v + 1) * gy)
            const int vrow = multi_view ? (int)(o_id / (uint32_t)Pview) * gy : 0;
            tiles[k] = (uint32_t)((vrow + ty) * gx + tx);
            vals[k] = o_id;
        }
    }
    R2_TS_AT(geom, 5);
}

// Same emission from the sorted records of the hinted depth order ({id, inclusive offset, tile rectangle} per sorted position,
// depth_order.hip): ONE coalesced load per lane instead of the order -> radii / record gathers (13.6 -> 8.8 us at 300k / 512^2).
__global__ void __launch_bounds__(256) raster_duplicate_sorted_kernel(
    int Pview, const uint4 *__restrict__ sorted, int gy, uint32_t *__restrict__ first, uint32_t *__restrict__ tiles,
    uint32_t *__restrict__ vals, const uint32_t *__restrict__ nvis, int gx, bool multi_view)
{
    R2_TS_AT(geom, 4);
    const int P = (int)*nvis;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave_first = j - lane;
    if (wave_first >= P) return;
    const uint4 r = sorted[min(j, P - 1)];
    const uint32_t id = r.x;
    const int rw = (int)((r.z >> 16) & 0xFFu) + 1;
    const uint32_t incl = r.y, excl = j < P ? r.y - depth_rect_count(r.z) : r.y;
    const int x0 = (int)(r.z & 0xFFu), y0 = (int)((r.z >> 8) & 0xFFu);
    if (j < P) first[id] = excl;   // where this Gaussian's instance run starts: the backward's scratch rows
    const uint32_t wbeg = __shfl(excl, 0);
    const int last_lane = min(63, P - 1 - wave_first);
    const uint32_t wend = __shfl(incl, last_lane);
    for (uint32_t base = wbeg; base < wend; base += 64) {
        const uint32_t k = base + lane;
        int lo = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {
            const int probe = lo + step;
            const uint32_t e = __shfl(excl, probe & 63);
            if (probe <= last_lane && e <= k) lo = probe;
        }
        const uint32_t o_excl = __shfl(excl, lo);
        const int o_x0 = __shfl(x0, lo), o_y0 = __shfl(y0, lo), o_rw = __shfl(rw, lo);
        const uint32_t o_id = __shfl(id, lo);
        if (k < wend) {
            const uint32_t local = k - o_excl;
            const int ty = o_y0 + (int)(local / (uint32_t)o_rw);
            const int tx = o_x0 + (int)(local % (uint32_t)o_rw);
            const int vrow = multi_view ? (int)(o_id / (uint32_t)Pview) * gy : 0;
            tiles[k] = (uint32_t)((vrow + ty) * gx + tx);
            vals[k] = o_id;
        }
    }
    R2_TS_AT(geom, 5);
}

// Emission BY OUTPUT RANGE, fused with the tile sort's histogram pass (its rs_upsweep): workgroup t produces exactly the
// keys of sort tile t -- instances [t * tile_keys, (t + 1) * tile_keys) -- and therefore also that tile's digit histogram, so
// the sort starts at its scan: no second pass over keys that were just written, one launch less.  The Gaussians of a range are
// the sorted records between the owners of its first instance and of the next range's first (depth_order_granule_owners, left by
// the ranking kernel: no search); their records are staged in LDS, every instance slot finds its owner by a mark + running-max
// scan over the range (owners start at increasing slots), then tile = rectangle origin + (row, column) of the slot inside
// the Gaussian's run.  Keys and ids are written coalesced, the histogram counts them on the way (LDS atomics).
constexpr int EMIT_THREADS = 512;
__global__ void __launch_bounds__(EMIT_THREADS) raster_emit_hist_kernel(
    const uint4 *__restrict__ sorted, const uint32_t *__restrict__ owners, const uint32_t *__restrict__ nvis_p, uint32_t R,
    uint32_t tile_keys, int bits, int Pview, int gx, int gy, bool multi_view, uint32_t *__restrict__ first,
    uint32_t *__restrict__ tiles, uint32_t *__restrict__ vals, uint32_t *__restrict__ H, uint32_t *__restrict__ clear_skip)
{
    extern __shared__ uint32_t emit_lds[];
    const uint32_t radix = 1u << bits;
    uint32_t *hist = emit_lds;                       // [radix]
    uint32_t *s_excl = hist + radix;                 // [tile_keys + 1]
    uint32_t *s_id = s_excl + tile_keys + 1;         // [tile_keys + 1]
    uint32_t *s_rect = s_id + tile_keys + 1;         // [tile_keys + 1]
    uint32_t *s_owner = s_rect + tile_keys + 1;      // [tile_keys] slot -> index into the staged records
    __shared__ uint32_t s_wmax[EMIT_THREADS / 64];
    R2_TS_AT(geom, 4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t k0 = blockIdx.x * tile_keys, k1 = min(k0 + tile_keys, R), nk = k1 - k0;
    const uint32_t ja = owners[k0 / TILE_SORT_GRANULE];
    const uint32_t jb = k1 < R ? owners[k1 / TILE_SORT_GRANULE] : *nvis_p - 1u;   // owner of the NEXT range's first instance (at most one
    const uint32_t nj = jb - ja + 1u;                                              // Gaussian too many: it marks no slot here)
    // the first rounds of records are requested before the LDS is cleared (a range holds ~tile_keys / 4 Gaussians)
    constexpr int PRE = 3;
    uint4 pre_r[PRE];
#pragma unroll
    for (int q = 0; q < PRE; ++q) {
        const uint32_t jj = (uint32_t)(q * EMIT_THREADS + tid);
        pre_r[q] = sorted[ja + min(jj, nj - 1u)];
    }
    if (clear_skip && blockIdx.x == 0 && tid == 0) clear_skip[0] = 0u;
    for (uint32_t d = tid; d < radix; d += EMIT_THREADS) hist[d] = 0u;
    for (uint32_t i = tid; i < nk; i += EMIT_THREADS) s_owner[i] = 0u;
    __syncthreads();
    auto stage = [&](uint32_t jj, const uint4 r) {
        const uint32_t excl = r.y - depth_rect_count(r.z);
        s_excl[jj] = excl;
        s_id[jj] = r.x;
        s_rect[jj] = r.z;
        if (excl >= k0 && excl < k1) {
            s_owner[excl - k0] = jj;
            first[r.x] = excl;   // where this Gaussian's instance run starts: the backward's scratch rows (written by the range
        }                        // its run STARTS in; the very first Gaussian starts in range 0)
    };
#pragma unroll
    for (int q = 0; q < PRE; ++q) {
        const uint32_t jj = (uint32_t)(q * EMIT_THREADS + tid);
        if (jj < nj) stage(jj, pre_r[q]);
    }
    for (uint32_t jj = (uint32_t)(PRE * EMIT_THREADS + tid); jj < nj; jj += EMIT_THREADS) stage(jj, sorted[ja + jj]);
    __syncthreads();
    // running maximum over the slots: thread t owns slots [t * ipt, (t + 1) * ipt)
    const uint32_t ipt = tile_keys / EMIT_THREADS;
    uint32_t run = 0u;
    for (uint32_t i = 0; i < ipt; ++i) {
        const uint32_t sidx = (uint32_t)tid * ipt + i;
        if (sidx < nk) run = max(run, s_owner[sidx]);
    }
    uint32_t incl = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if (lane >= d) incl = max(incl, up);
    }
    if (lane == 63) s_wmax[wave] = incl;
    __syncthreads();
    uint32_t pre = __shfl_up(incl, 1);
    if (lane == 0) pre = 0u;
    for (int w = 0; w < wave; ++w) pre = max(pre, s_wmax[w]);
    run = pre;
    for (uint32_t i = 0; i < ipt; ++i) {
        const uint32_t sidx = (uint32_t)tid * ipt + i;
        if (sidx < nk) {
            run = max(run, s_owner[sidx]);
            s_owner[sidx] = run;
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < nk; i += EMIT_THREADS) {
        const uint32_t o = s_owner[i];
        const uint32_t rect = s_rect[o], id = s_id[o];
        const uint32_t local = k0 + i - s_excl[o];
        const uint32_t rw = ((rect >> 16) & 0xFFu) + 1u;
        // local / rw without the integer division: local < 65536 and rw <= 256, so (local + 0.5) / rw stays 0.5 / rw away from
        // the integers -- far more than float rounding moves it
        const uint32_t row = (uint32_t)(((float)local + 0.5f) * __builtin_amdgcn_rcpf((float)rw)), col = local - row * rw;
        const int vrow = multi_view ? (int)(id / (uint32_t)Pview) * gy : 0;
        const uint32_t tile = (uint32_t)((vrow + (int)((rect >> 8) & 0xFFu) + (int)row) * gx + (int)(rect & 0xFFu) + (int)col);
        tiles[k0 + i] = tile;
        vals[k0 + i] = id;
        atomicAdd(&hist[tile & (radix - 1u)], 1u);
    }
    __syncthreads();
    uint32_t *__restrict__ hrow = H + (size_t)blockIdx.x * radix;
    for (uint32_t d = tid; d < radix; d += EMIT_THREADS) hrow[d] = hist[d];
    R2_TS_AT(geom, 5);
}

// ------------------------------------------------------------------ backward: fused geometry gradient
// One pass per Gaussian:
//   1. reduce the per-instance moment rows written by the render backward (this Gaussian's instances are
//      contiguous in the UNSORTED list: [first, first + tiles_touched)), in a fixed order -> deterministic,
//      atomic-free gradients (the reference accumulates with float atomicAdd, RAS/backward.cu:562-572);
//   2. turn the moments into dL/dmean2D, dL/dconic, dL/dopacity, dL/dmu (the 7 sums of the reference);
//   3. computeCov2DCUDA (RAS/backward.cu:145-330) + preprocessCUDA backward (RAS/backward.cu:402-444).
// Outputs are ASSIGNED; the caller's zero-initialisation covers the rows of culled Gaussians.
constexpr int GB_ROWS = 256;   // moment rows a wave streams per pass (4 per lane)
// (Measured and left out, round 5: __launch_bounds__(256, 4) makes hipcc allocate 94 instead of 98 VGPRs, i.e. five waves per SIMD by
// registers -- the grid of 1172 workgroups is 1.14 rounds of the 1024 slots that four allow, and the stamps show the second round
// starting 12 us after the first.  26.2 us against 25.2: the LDS (32 KB per workgroup) still stops at four per CU, and with 192 or 128
// rows per pass -- 24 / 16 KB, six or more per CU -- the kernel takes 27.7 us: the row passes get shorter, the round trips do not.)
template <bool MV>
__global__ void __launch_bounds__(256) raster_geom_backward_kernel(
    int P, int Vn, const float *__restrict__ means3D, const int *__restrict__ radii, const float *__restrict__ cov3Ds,
    int cov_per_view, const float *__restrict__ scales, const float *__restrict__ rotations, float scale_modifier,
    float h_x, float h_y, float tan_fovx, float tan_fovy, const float *__restrict__ views,
    const float *__restrict__ projs, const float4 *__restrict__ rec, const float2 *__restrict__ op_mu,
    const uint32_t *__restrict__ first_inst,
    const uint32_t *__restrict__ tiles_touched, const float4 *__restrict__ part, float W_half, float H_half,
    float *__restrict__ dL_dconics, float *__restrict__ dL_dmus, float *__restrict__ dL_dmean2D,
    float *__restrict__ dL_dopacity, float *__restrict__ dL_dmeans, float *__restrict__ dL_dcov,
    float *__restrict__ dL_dscale, float *__restrict__ dL_drot, int mode)
{
    // one lane per Gaussian; its V view instances g = v * P + idx are visited in view order and their gradients summed in
    // that order (deterministic).  Per view: dL/dmean2D, dL/dconic, dL/dmu (rows g); summed over the views: dL/dopacity,
    // dL/dmean3D, dL/dcov3D and, through ONE covariance backward on the sum (it is linear), dL/dscale, dL/drot (rows idx).
    // (lanes past the end stay in the kernel: the row streaming below is a wave-cooperative step; they work on a clamped
    // index and store nothing)
    const int idx_raw = blockIdx.x * 256 + threadIdx.x;
    const bool in_range = idx_raw < P;
    const int idx = in_range ? idx_raw : P - 1;
    const int lane = threadIdx.x & 63;
    __shared__ float4 s_rows[256 / 64][GB_ROWS * 2];   // per wave: one pass of moment rows (32 bytes each)
    float4 *const wrows = s_rows[threadIdx.x >> 6];
    R2_TS_AT(geom, 6);
    const int V = MV ? Vn : 1;   // single view: a compile-time trip count (the loop folds away)
    const float3 mean = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
    float sc_in[3] = { 0.f, 0.f, 0.f };
    float4 rot_in = make_float4(0.f, 0.f, 0.f, 0.f);
    if (scales != nullptr) {
        sc_in[0] = scales[3 * idx]; sc_in[1] = scales[3 * idx + 1]; sc_in[2] = scales[3 * idx + 2];
        rot_in = reinterpret_cast<const float4 *>(rotations)[idx];
    }
    float acc_op = 0.f, acc_mean[3] = { 0.f, 0.f, 0.f }, acc_cov[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
    for (int v = 0; v < V; ++v) {
    const int g = v * P + idx;
    const float *__restrict__ view = views + 16 * v;
    const float *__restrict__ proj = projs + 16 * v;
    const bool live = in_range && radii[g] > 0;

    // ---- 1. moments of w = G * dL/dpix over all tiles of this view instance
    // every per-instance input is requested up front so that its latency overlaps the (dependent) row gathers
    const float4 ra = rec[2 * g], rb = rec[2 * g + 1];
    const uint32_t first = first_inst[g], ninst = live ? tiles_touched[g] : 0u;
    const float2 om = op_mu[g];
    float cov3D[6];
    // the covariance comes from the forward's state (recomputing it from scales / rotations instead -- 24 B/Gaussian less
    // to write and read -- was measured: the preprocess got no faster, this kernel 1.2 us slower: both are latency-bound)
    const size_t cbase = cov_per_view ? (size_t)g : (size_t)idx;   // cov3D_precomp is per Gaussian, the state per view instance
#pragma unroll
    for (int k = 0; k < 6; ++k) cov3D[k] = cov3Ds[6 * cbase + k];
    float S0 = 0.f, S1 = 0.f, S2 = 0.f, S3 = 0.f, S4 = 0.f, S5 = 0.f;
#ifdef R2_EXP_TS
    if (ninst == 0xFFFFFFFFu) S0 = cov3D[0] + ra.x + om.x;   // keep the loads ahead of the stamp
    R2_TS_AT(geom, 8);
#endif
    // This Gaussian's rows are contiguous (the render backward stores each instance's row at its EMISSION index), but a lane
    // that walks its own run makes its wave wait one gather round trip per step of its WIDEST Gaussian: stamped inside the
    // kernel, that loop was 11 of the 16 us a workgroup lives.  So the wave streams the rows of its 64 Gaussians together:
    // the concatenated runs are dealt out 64 * 4 rows at a time, one row per lane and load (owner = last lane whose exclusive
    // offset is <= k, found like in the duplicate kernel), parked in LDS, and every lane then adds ITS rows from LDS in row
    // order -- the same sums in the same order as before (bit-reproducible, bit-identical to the per-lane walk), one gather
    // round trip per 256 rows instead of one per 4 rows of the widest lane.
    {
        uint32_t incl = ninst;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        const uint32_t excl = incl - ninst;
        const uint32_t total = __shfl(incl, 63);
        for (uint32_t base = 0; base < total; base += GB_ROWS) {   // wave-uniform
            float4 m0[GB_ROWS / 64], m1[GB_ROWS / 64];
#pragma unroll
            for (int u = 0; u < GB_ROWS / 64; ++u) {
                const uint32_t k = min(base + (uint32_t)(u * 64 + lane), total - 1u);   // clamped: branch-free loads
                int lo = 0;
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) {
                    const int probe = lo + step;
                    const uint32_t e = __shfl(excl, probe & 63);
                    if (probe <= 63 && e <= k) lo = probe;
                }
                const size_t row = (size_t)__shfl(first, lo) + (k - __shfl(excl, lo));
                m0[u] = part[2 * row];
                m1[u] = part[2 * row + 1];
            }
            __builtin_amdgcn_wave_barrier();   // the previous pass has been read
#pragma unroll
            for (int u = 0; u < GB_ROWS / 64; ++u) {
                wrows[2 * (u * 64 + lane)] = m0[u];
                wrows[2 * (u * 64 + lane) + 1] = m1[u];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t r0 = max(excl, base), r1 = min(excl + ninst, base + (uint32_t)GB_ROWS);
            for (uint32_t r = r0; r < r1; ++r) {
                const float4 a0 = wrows[2 * (r - base)], a1 = wrows[2 * (r - base) + 1];
                S0 += a0.x; S1 += a0.y; S2 += a0.z; S3 += a0.w; S4 += a1.x; S5 += a1.y;
            }
        }
    }
    if (!live) {
        if (in_range) {
            // culled in this view: all-zero per-view rows.  The reference gets them from the torch boundary's zero-filled
            // tensors (SUB/rasterize_points.cu:124-131); writing them here spares the caller a memset.
            dL_dmean2D[3 * g + 0] = 0.f; dL_dmean2D[3 * g + 1] = 0.f; dL_dmean2D[3 * g + 2] = 0.f;
            dL_dmus[g] = 0.f;
            reinterpret_cast<float4 *>(dL_dconics)[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        continue;
    }
    R2_TS_AT(geom, 9);
    // ---- 2. the reference's accumulated sums (RAS/backward.cu:556-572), conic un-scaled from log2e units
    const float op = om.x, mu_f = om.y, opmu = op * mu_f;
    const float cA = ra.z * (-2.0f * LN2), cB = ra.w * (-LN2), cC = rb.x * (-2.0f * LN2);
    const float g2x = opmu * W_half * (-cA * S1 - cB * S2);
    const float g2y = opmu * H_half * (-cC * S2 - cB * S1);
    const float gx_ = -0.5f * opmu * S3, gy_ = -opmu * S4, gz_ = -0.5f * opmu * S5;
    const float dL_dmu = op * S0;
    dL_dmean2D[3 * g + 0] = g2x;
    dL_dmean2D[3 * g + 1] = g2y;
    dL_dmean2D[3 * g + 2] = 0.f;   // RAS/backward.cu never touches the third component
    acc_op += mu_f * S0;
    dL_dmus[g] = dL_dmu;
    reinterpret_cast<float4 *>(dL_dconics)[g] = make_float4(gx_, gy_, 0.f, gz_);

    // ---- 3. geometry chain
    Cov2D c;
    cov2d_common(mean, h_x, h_y, tan_fovx, tan_fovy, cov3D, view, mode, c);
    const M3 &M = c.M;
    const M3 &Wm = c.W;
    const float hata = c.cov.m[0][0], hatb = c.cov.m[0][1], hatc = c.cov.m[0][2];
    const float hatd = c.cov.m[1][1], hate = c.cov.m[1][2], hatf = c.cov.m[2][2];

    float da = 0, db = 0, dc = 0, dd = 0, de = 0, df = 0;
    const float denom = hata * hatd - hatb * hatb;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    const float diamond = hata * hatd - hatb * hatb;
    const float circ = hata * hatd * hatf + 2 * hatb * hatc * hate - hata * hate * hate - hatf * hatb * hatb - hatd * hatc * hatc;
    const float mu = mu_of(circ, diamond);
    const float pi_mu = (float)(M_PI / (double)(mu + 0.0000001f));
    const float circ_diamond = circ / diamond;

    float o[6] = { 0, 0, 0, 0, 0, 0 };
    if (denom2inv != 0.0f && mu != 0.0f) {
        da = denom2inv * (-hatd * hatd * gx_ + hatb * hatd * gy_ + (denom - hata * hatd) * gz_);
        dd = denom2inv * (-hata * hata * gz_ + hata * hatb * gy_ + (denom - hata * hatd) * gx_);
        db = denom2inv * (2 * hatb * hatd * gx_ - (denom + 2 * hatb * hatb) * gy_ + 2 * hata * hatb * gz_);

        da += pi_mu * ((hatd * hatf - hate * hate) / diamond - hatd * circ_diamond / diamond) * dL_dmu;
        db += pi_mu * ((2 * hatc * hate - 2 * hatf * hatb) / diamond + 2 * hatb * circ_diamond / diamond) * dL_dmu;
        dc += pi_mu * ((2 * hatb * hate - 2 * hatd * hatc) / diamond) * dL_dmu;
        dd += pi_mu * ((hata * hatf - hatc * hatc) / diamond - hata * circ_diamond / diamond) * dL_dmu;
        de += pi_mu * ((2 * hatb * hatc - 2 * hata * hate) / diamond) * dL_dmu;
        df += pi_mu * ((hata * hatd - hatb * hatb) / diamond) * dL_dmu;
        dcov_from_dhat(M, da, db, dc, dd, de, df, o);
    }
    // else: all six stay 0 (Q8)
#pragma unroll
    for (int k = 0; k < 6; ++k) acc_cov[k] += o[k];

    float3 gmean = make_float3(0.f, 0.f, 0.f);
    if (mode == 1) {
#define MM(c_, r_) (M.m[c_][r_])
#define WW(c_, r_) (Wm.m[c_][r_])
        const float a = cov3D[0], b = cov3D[1], cc = cov3D[2], d = cov3D[3], e = cov3D[4], f = cov3D[5];
        const float dM00 = 2*(MM(0,0)*a+MM(0,1)*b + MM(0,2)*cc)*da + (MM(1,0)*a+MM(1,1)*b+MM(1,2)*cc)*db + (MM(2,0)*a+MM(2,1)*b+MM(2,2)*cc)*dc;
        const float dM01 = 2*(MM(0,0)*b+MM(0,1)*d + MM(0,2)*e)*da + (MM(1,0)*b+MM(1,1)*d+MM(1,2)*e)*db + (MM(2,0)*b+MM(2,1)*d+MM(2,2)*e)*dc;
        const float dM02 = 2*(MM(0,0)*cc+MM(0,1)*e + MM(0,2)*f)*da + (MM(1,0)*cc+MM(1,1)*e+MM(1,2)*f)*db + (MM(2,0)*cc+MM(2,1)*e+MM(2,2)*f)*dc;
        const float dM10 = (MM(0,0)*a+MM(0,1)*b+MM(0,2)*cc)*db + 2*(MM(1,0)*a+MM(1,1)*b+MM(1,2)*cc)*dd + (MM(2,0)*a+MM(2,1)*b+MM(2,2)*cc)*de;
        const float dM11 = (MM(0,0)*b+MM(0,1)*d+MM(0,2)*e)*db + 2*(MM(1,0)*b+MM(1,1)*d+MM(1,2)*e)*dd + (MM(2,0)*b+MM(2,1)*d+MM(2,2)*e)*de;
        const float dM12 = (MM(0,0)*cc+MM(0,1)*e+MM(0,2)*f)*db + 2*(MM(1,0)*cc+MM(1,1)*e+MM(1,2)*f)*dd + (MM(2,0)*cc+MM(2,1)*e+MM(2,2)*f)*de;
        const float dM20 = (MM(0,0)*a+MM(0,1)*b+MM(0,2)*cc)*dc + (MM(1,0)*a+MM(1,1)*b+MM(1,2)*cc)*de + 2*(MM(2,0)*a+MM(2,1)*b+MM(2,2)*cc)*df;
        const float dM21 = (MM(0,0)*b+MM(0,1)*d+MM(0,2)*e)*dc + (MM(1,0)*b+MM(1,1)*d+MM(1,2)*e)*de + 2*(MM(2,0)*b+MM(2,1)*d+MM(2,2)*e)*df;
        const float dM22 = (MM(0,0)*cc+MM(0,1)*e+MM(0,2)*f)*dc + (MM(1,0)*cc+MM(1,1)*e+MM(1,2)*f)*de + 2*(MM(2,0)*cc+MM(2,1)*e+MM(2,2)*f)*df;

        const float dJ00 = WW(0,0)*dM00 + WW(0,1)*dM01 + WW(0,2)*dM02;
        const float dJ02 = WW(2,0)*dM00 + WW(2,1)*dM01 + WW(2,2)*dM02;
        const float dJ11 = WW(1,0)*dM10 + WW(1,1)*dM11 + WW(1,2)*dM12;
        const float dJ12 = WW(2,0)*dM10 + WW(2,1)*dM11 + WW(2,2)*dM12;
        const float dJ20 = WW(0,0)*dM20 + WW(0,1)*dM21 + WW(0,2)*dM22;
        const float dJ21 = WW(1,0)*dM20 + WW(1,1)*dM21 + WW(1,2)*dM22;
        const float dJ22 = WW(2,0)*dM20 + WW(2,1)*dM21 + WW(2,2)*dM22;
#undef MM
#undef WW
        const float tx = c.tx, ty = c.ty, tz = c.tz;
        const float inv_tz = 1.f / tz;
        const float inv_tz2 = inv_tz * inv_tz;
        const float inv_tz3 = inv_tz2 * inv_tz;
        const float cc0 = sqrtf(tx * tx + ty * ty + tz * tz);
        const float icc3 = 1 / (cc0 * cc0 * cc0);
        const float dtx = c.xmul * (-h_x*inv_tz2*dJ02 + (1/cc0 - tx*tx*icc3)*dJ20 - tx*ty*icc3*dJ21 - tx*tz*icc3*dJ22);
        const float dty = c.ymul * (-h_y*inv_tz2*dJ12 - tx*ty*icc3*dJ20 + (1/cc0 - ty*ty*icc3)*dJ21 - ty*tz*icc3*dJ22);
        const float dtz = -h_x*inv_tz2*dJ00 + 2*h_x*tx*inv_tz3*dJ02 - h_y*inv_tz2*dJ11 + 2*h_y*ty*inv_tz3*dJ12 - tx*tz*icc3*dJ20 - ty*tz*icc3*dJ21 + (1/cc0-tz*tz*icc3)*dJ22;
        gmean.x = view[0] * dtx + view[1] * dty + view[2] * dtz;
        gmean.y = view[4] * dtx + view[5] * dty + view[6] * dtz;
        gmean.z = view[8] * dtx + view[9] * dty + view[10] * dtz;
    }

    // mean gradient through the perspective divide (RAS/backward.cu:419-437)
    const float4 m_hom = xform4x4(mean, proj);
    const float m_w = 1.0f / (m_hom.w + 0.0000001f);
    const float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
    const float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
    const float g0 = g2x, g1 = g2y;
    const float ddx = (proj[0] * m_w - proj[3] * mul1) * g0 + (proj[1] * m_w - proj[3] * mul2) * g1;
    const float ddy = (proj[4] * m_w - proj[7] * mul1) * g0 + (proj[5] * m_w - proj[7] * mul2) * g1;
    const float ddz = (proj[8] * m_w - proj[11] * mul1) * g0 + (proj[9] * m_w - proj[11] * mul2) * g1;
    acc_mean[0] += gmean.x + ddx;
    acc_mean[1] += gmean.y + ddy;
    acc_mean[2] += gmean.z + ddz;
    }   // views
    if (!in_range) return;

    dL_dopacity[idx] = acc_op;
    dL_dmeans[3 * idx + 0] = acc_mean[0];
    dL_dmeans[3 * idx + 1] = acc_mean[1];
    dL_dmeans[3 * idx + 2] = acc_mean[2];
#pragma unroll
    for (int k = 0; k < 6; ++k) dL_dcov[6 * idx + k] = acc_cov[k];
    if (scales != nullptr) {
        float ds[3];
        float4 dq;
        cov3d_backward(sc_in[0], sc_in[1], sc_in[2], scale_modifier, rot_in, acc_cov, ds, &dq);
        dL_dscale[3 * idx + 0] = ds[0];
        dL_dscale[3 * idx + 1] = ds[1];
        dL_dscale[3 * idx + 2] = ds[2];
        reinterpret_cast<float4 *>(dL_drot)[idx] = dq;
    } else {   // cov3D_precomp path: no scale / rotation gradient
        if (dL_dscale) { dL_dscale[3 * idx + 0] = 0.f; dL_dscale[3 * idx + 1] = 0.f; dL_dscale[3 * idx + 2] = 0.f; }
        if (dL_drot) reinterpret_cast<float4 *>(dL_drot)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    R2_TS_AT(geom, 7);
}

// ------------------------------------------------------------------ host launchers
int launch_raster_preprocess(const RasterGeom &g, int P, int V, const float *means3D, const float *scales, float scale_modifier,
                             const float *rotations, const float *opacities, const float *cov3D_precomp,
                             const float *views, const float *projs, int W, int H, float tan_fovx, float tan_fovy,
                             int mode, int *radii, uint32_t *thin_flag, const DepthReg &reg, bool store_cov3D, hipStream_t s)
{
    const float focal_y = H / (2.0f * tan_fovy);
    const float focal_x = W / (2.0f * tan_fovx);
    const int gx = (W + TILE2D - 1) / TILE2D, gy = (H + TILE2D - 1) / TILE2D;
    raster_preprocess_kernel<<<dim3((unsigned)(((size_t)P * V + 255) / 256)), dim3(256), 0, s>>>(
        P, V, means3D, scales, scale_modifier, rotations, opacities, cov3D_precomp, views, projs, W, H, tan_fovx, tan_fovy,
        focal_x, focal_y, mode, gx, gy, radii, g.rec, g.depth_key, store_cov3D ? g.cov3D : nullptr, g.tiles_touched, g.op_mu, thin_flag,
        reg);
    return 0;
}

int launch_raster_preprocess_tf(const RasterGeom &g, int P, int V, const TFGrid &grid, const TFSlabs &slabs, const float *means3D, const float *scales,
                                float scale_modifier, const float *rotations, const float *opacities, const float *cov3D_precomp,
                                const float *view, const float *proj, int W, int H, float tan_fovx, float tan_fovy, int mode,
                                int *radii, TFCounters *ctr, hipStream_t s)
{
    const float focal_y = H / (2.0f * tan_fovy);
    const float focal_x = W / (2.0f * tan_fovx);
    const int gx = (W + TILE2D - 1) / TILE2D, gy = (H + TILE2D - 1) / TILE2D;
    bool share = (int)grid.wgs > device_cu_count() && grid.threads > TF_THREADS_MAX / 2;
    {   // stacked views: the view arithmetic does not fit the 64-register instantiation (35 spilled registers); one workgroup per CU
        // at a time measured 1.5 % (V = 2) and 0.7 % (V = 4) faster per call than two spilling ones (R2_TF_MV_SHARE=1: the other way)
        static const int mv_share = [] { const char *e = getenv("R2_TF_MV_SHARE"); return e ? atoi(e) : 0; }();
        if (V > 1 && !mv_share) share = false;
    }
#define R2_TF_PRE(SLB, SHR, MVW)                                                                                                       \
    raster_preprocess_tf_kernel<SLB, SHR, MVW><<<dim3(grid.wgs), dim3(grid.threads), (size_t)gx * gy * V * slabs.n * sizeof(uint32_t), s>>>( \
        P, V, grid.per_wg, slabs, means3D, scales, scale_modifier, rotations, opacities, cov3D_precomp, view, proj, W, H, tan_fovx, tan_fovy, \
        focal_x, focal_y, mode, gx, gy, radii, g.rec, g.depth_key, g.cov3D, g.tiles_touched, g.op_mu, g.first, g.tf_rect, g.tf_wgoff,      \
        g.tf_wgmm, ctr)
#define R2_TF_PRE2(SLB)                                                                                                                \
    if (share) { if (V > 1) R2_TF_PRE(SLB, true, true); else R2_TF_PRE(SLB, true, false); }                                             \
    else { if (V > 1) R2_TF_PRE(SLB, false, true); else R2_TF_PRE(SLB, false, false); }
    if (slabs.n > 1u) { R2_TF_PRE2(true) }
    else { R2_TF_PRE2(false) }
#undef R2_TF_PRE2
#undef R2_TF_PRE
    return 0;
}

int launch_raster_duplicate(const RasterGeom &g, const RasterBinning &b, int P, int V, const int *radii, int W, int H,
                            const uint32_t *nvis, hipStream_t s)
{
    const int gx = (W + TILE2D - 1) / TILE2D, gy = (H + TILE2D - 1) / TILE2D;
    const int PV = P * V;
    raster_duplicate_kernel<<<dim3((PV + 255) / 256), dim3(256), 0, s>>>(PV, P, g.rec, g.order, g.offsets, radii, gx, gy,
                                                                         g.first, b.tiles_unsorted, b.vals_unsorted, nvis);
    return 0;
}

int launch_raster_duplicate_sorted(const RasterGeom &g, const RasterBinning &b, int P, int V, int W, int H, const uint32_t *nvis,
                                   hipStream_t s)
{
    const int gx = (W + TILE2D - 1) / TILE2D, gy = (H + TILE2D - 1) / TILE2D;
    const int PV = P * V;
    raster_duplicate_sorted_kernel<<<dim3((PV + 255) / 256), dim3(256), 0, s>>>(P, depth_order_sorted_records(g.dorder_temp, (size_t)PV), gy,
                                                                                g.first, b.tiles_unsorted, b.vals_unsorted, nvis, gx, V > 1);
    return 0;
}

bool launch_raster_emit_hist(const RasterGeom &g, const RasterBinning &b, int P, int V, int W, int H, const uint32_t *nvis, size_t R,
                             const TileSortPlan &plan, hipStream_t s)
{
    const int gx = (W + TILE2D - 1) / TILE2D, gy = (H + TILE2D - 1) / TILE2D;
    const int PV = P * V;
    uint32_t cap = 0;
    const uint32_t *owners = depth_order_granule_owners(g.dorder_temp, (size_t)PV, &cap);
    if (R / TILE_SORT_GRANULE + 1 > cap || plan.tile_keys % EMIT_THREADS != 0) return false;   // (a cloud of huge Gaussians)
    const size_t lds = ((size_t)(1u << plan.bits) + 3 * ((size_t)plan.tile_keys + 1) + plan.tile_keys) * sizeof(uint32_t);
    static signed char lds_state[R2_MAX_DEVICES] = {};
    const bool attr_ok = allow_dynamic_lds(reinterpret_cast<const void *>(raster_emit_hist_kernel), 150 * 1024, lds_state);
    if (!attr_ok || lds > 150 * 1024 || lds > device_lds_optin_bytes()) return false;
    raster_emit_hist_kernel<<<dim3(plan.ntiles), dim3(EMIT_THREADS), lds, s>>>(
        depth_order_sorted_records(g.dorder_temp, (size_t)PV), owners, nvis, (uint32_t)R, plan.tile_keys, plan.bits, P, gx, gy, V > 1,
        g.first, b.tiles_unsorted, b.vals_unsorted, plan.H, plan.skip);
    // a refused launch (the attribute was accepted but the configuration is not) must not count as "histograms ready": the
    // caller then emits with the plain kernel and runs the sort's own upsweep
    if (hipGetLastError() != hipSuccess) return false;
    return true;
}

int launch_mark_visible(int P, const float *means3D, const float *view, uint8_t *present, hipStream_t s)
{
    mark_visible_kernel<<<dim3((P + 255) / 256), dim3(256), 0, s>>>(P, means3D, view, present);
    return 0;
}

int launch_raster_geom_backward(int P, int V, const float *means3D, const int *radii, const float *cov3D, const float *scales,
                                const float *rotations, float scale_modifier, int W, int H, float tan_fovx,
                                float tan_fovy, const float *views, const float *projs, float *dL_dconic,
                                float *dL_dmu, float *dL_dmean2D, float *dL_dopacity, float *dL_dmean3D,
                                float *dL_dcov3D, float *dL_dscale, float *dL_drot, int mode, const RasterGeom &g,
                                const float *part, hipStream_t s)
{
    const float h_y = H / (2.0f * tan_fovy);
    const float h_x = W / (2.0f * tan_fovx);
    // cov3D == the state's array: one covariance per view instance; a caller's cov3D_precomp: one per Gaussian
    if (V > 1)
        raster_geom_backward_kernel<true><<<dim3((P + 255) / 256), dim3(256), 0, s>>>(
            P, V, means3D, radii, cov3D, cov3D == g.cov3D ? 1 : 0, scales, rotations, scale_modifier, h_x, h_y, tan_fovx, tan_fovy,
            views, projs, g.rec, g.op_mu, g.first, g.tiles_touched, reinterpret_cast<const float4 *>(part), 0.5f * (float)W,
            0.5f * (float)H, dL_dconic, dL_dmu, dL_dmean2D, dL_dopacity, dL_dmean3D, dL_dcov3D, dL_dscale, dL_drot, mode);
    else
        raster_geom_backward_kernel<false><<<dim3((P + 255) / 256), dim3(256), 0, s>>>(
            P, V, means3D, radii, cov3D, cov3D == g.cov3D ? 1 : 0, scales, rotations, scale_modifier, h_x, h_y, tan_fovx, tan_fovy,
            views, projs, g.rec, g.op_mu, g.first, g.tiles_touched, reinterpret_cast<const float4 *>(part), 0.5f * (float)W,
            0.5f * (float)H, dL_dconic, dL_dmu, dL_dmean2D, dL_dopacity, dL_dmean3D, dL_dcov3D, dL_dscale, dL_drot, mode);
    return 0;
}

}  // namespace r2
