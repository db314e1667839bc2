// voxel_geom.hip -- per-Gaussian kernels of the 3D voxelizer: voxel-space inverse covariance + tile cube
// (forward), key/value emission, fused geometry backward.  Compiled with -ffp-contract=off (r2_math.hpp).
//
// Reference: VOX/forward.cu:58-178, VOX/auxiliary.h:27-39, VOX/voxelizer_impl.cu:54-101,
//            VOX/backward.cu:86-213.
#include "r2_math.hpp"
#include "voxel_state.hpp"

namespace r2 {

// tile cube of a box of half-widths rad around p (VOX/auxiliary.h:27-39), float->int truncation
__device__ __forceinline__ void tile_cube(float3 p, float3 rad, int gx, int gy, int gz, int3 &lo, int3 &hi)
{
    lo.x = min(gx, max(0, (int)((p.x - rad.x) / TILE3D)));
    lo.y = min(gy, max(0, (int)((p.y - rad.y) / TILE3D)));
    lo.z = min(gz, max(0, (int)((p.z - rad.z) / TILE3D)));
    hi.x = min(gx, max(0, (int)((p.x + rad.x + TILE3D - 1) / TILE3D)));
    hi.y = min(gy, max(0, (int)((p.y + rad.y + TILE3D - 1) / TILE3D)));
    hi.z = min(gz, max(0, (int)((p.z + rad.z + TILE3D - 1) / TILE3D)));
}

// cov = M^T Vrk^T M with M = diag(1/dVoxel)  (VOX/forward.cu:110-118)
__device__ __forceinline__ void voxel_cov(const float *cov3D, float dvx, float dvy, float dvz, M3 &M, float *h)
{
    const M3 Vrk = m3(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    M = m3(1.f / dvx, 0.0f, 0.0f, 0.0f, 1.f / dvy, 0.0f, 0.0f, 0.0f, 1.f / dvz);
    const M3 cov = mul(mul(tr(M), tr(Vrk)), M);
    h[0] = cov.m[0][0]; h[1] = cov.m[0][1]; h[2] = cov.m[0][2];
    h[3] = cov.m[1][1]; h[4] = cov.m[1][2]; h[5] = cov.m[2][2];
}

// Inverse of the voxel-space covariance (VOX/forward.cu:110-135), upper triangle; false: singular (the reference then returns
// with radii = 0).
__device__ __forceinline__ bool voxel_inverse(const float *cov3D, float dvx, float dvy, float dvz, float *inv)
{
    M3 M;
    float h[6];
    voxel_cov(cov3D, dvx, dvy, dvz, M, h);
    const float a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5];
    const float det = a * d * f + 2 * b * c * e - a * e * e - f * b * b - d * c * c;
    if (det == 0.0f) return false;
    const float det_inv = 1.f / det;
    inv[0] = (d * f - e * e) * det_inv;
    inv[1] = (c * e - b * f) * det_inv;
    inv[2] = (b * e - c * d) * det_inv;
    inv[3] = (a * f - c * c) * det_inv;
    inv[4] = (b * c - a * e) * det_inv;
    inv[5] = (a * d - b * b) * det_inv;
    return true;
}
__device__ __forceinline__ float3 voxel_position(float3 p, const VoxelGrid &v, float dvx, float dvy, float dvz)
{
    return make_float3((p.x - v.cx + v.sx / 2) / dvx, (p.y - v.cy + v.sy / 2) / dvy, (p.z - v.cz + v.sz / 2) / dvz);
}
// An x-slab call (r2_voxel_forward_slab) runs the FULL grid's arithmetic -- (sx, cx, fnx, fgx) are the full volume's, so positions,
// radii and the tile cube come out bit for bit as in the unsharded call -- and then keeps the slab's tile layers only, renumbered from
// 0: everything downstream (emission, lists, ranges, the backward's emission index) sees a grid of v.gx x gy x gz tiles.  For an
// ordinary call ox = 0 and gx = fgx: nothing is clipped.
__device__ __forceinline__ void slab_clip(const VoxelGrid &v, int3 &lo, int3 &hi)
{
    const int t0 = v.ox / TILE3D;
    lo.x = min(max(lo.x, t0), t0 + v.gx) - t0;
    hi.x = min(max(hi.x, t0), t0 + v.gx) - t0;
}

// The preprocess in two parts (one after the other in voxel_preprocess_kernel; as two kernels in the stick-first chain,
// voxel_sticks.hip, where the second one runs while the host sizes the binning state).
// Part 1, everything the BINNING needs: radii, tiles_touched, the depth key, the tile cube (and the 3D covariance).
// -> the Gaussian's number of tiles (0: it emits nothing); pv / inv: voxel-space position and inverse covariance for part 2.
// EARLY_CULL (small grids, where ~98 % of the Gaussians miss the volume): the volume / tile-cube tests come first, so a culled
// Gaussian costs two loads and never computes or stores its covariance.  Same outcome: the reference returns with radii = 0 on
// whichever test fails first (VOX/forward.cu:120-160) and nothing of a culled Gaussian's state is read again.
template <bool EARLY_CULL = false>
__device__ __forceinline__ uint32_t voxel_cull_one(
    int idx, const float *__restrict__ means3D, const float *__restrict__ scales, float scale_modifier,
    const float *__restrict__ rotations, const float *__restrict__ cov3D_precomp, const VoxelGrid &v,
    int *__restrict__ radii_x, int *__restrict__ radii_y, int *__restrict__ radii_z, uint32_t *__restrict__ depth_key,
    float *__restrict__ cov3Ds, uint32_t *__restrict__ tiles_touched, uint4 *__restrict__ cube, float3 &pv_out, float *inv,
    int3 &lo, int3 &hi)
{
    radii_x[idx] = 0;
    radii_y[idx] = 0;
    radii_z[idx] = 0;
    tiles_touched[idx] = 0;
    const float dvx = v.sx / (float)v.fnx, dvy = v.sy / (float)v.ny, dvz = v.sz / (float)v.nz;
    const float3 p = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
    // low sort word of the reference: the raw bits of world z ("just give a value", VOX/forward.cu:166; as
    // unsigned ints, negative z sorts after positive -- quirk Q10).  Culled Gaussians emit nothing, so their
    // position in the depth order is irrelevant.
    depth_key[idx] = __float_as_uint(p.z);
    if (EARLY_CULL) {
        const float ms = fmaxf(fmaxf(scales[3 * idx], scales[3 * idx + 1]), scales[3 * idx + 2]);
        const float3 rd = make_float3(ceilf((3.f * ms) / dvx), ceilf((3.f * ms) / dvy), ceilf((3.f * ms) / dvz));
        const float3 q = voxel_position(p, v, dvx, dvy, dvz);
        if (q.x + rd.x < 0 || q.y + rd.y < 0 || q.z + rd.z < 0 || q.x - rd.x > (float)v.fnx || q.y - rd.y > (float)v.ny ||
            q.z - rd.z > (float)v.nz)
            return 0u;
        int3 l0, h0;
        tile_cube(q, rd, v.fgx, v.gy, v.gz, l0, h0);
        slab_clip(v, l0, h0);
        if ((h0.x - l0.x) * (h0.y - l0.y) * (h0.z - l0.z) == 0) return 0u;
    }

    float cov3D[6];
    if (cov3D_precomp != nullptr) {
#pragma unroll
        for (int k = 0; k < 6; ++k) cov3D[k] = cov3D_precomp[6 * idx + k];
    } else {
        const float4 q = reinterpret_cast<const float4 *>(rotations)[idx];
        cov3d_from_scale_rot(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2], scale_modifier, q, cov3D);
        if (cov3Ds != nullptr)
#pragma unroll
            for (int k = 0; k < 6; ++k) cov3Ds[6 * idx + k] = cov3D[k];
    }
    if (!voxel_inverse(cov3D, dvx, dvy, dvz, inv)) return 0u;

    // radius from the RAW scales, no scale_modifier (reference quirk Q5, VOX/forward.cu:137-143)
    const float max_scale = fmaxf(fmaxf(scales[3 * idx], scales[3 * idx + 1]), scales[3 * idx + 2]);
    const float3 rad = make_float3(ceilf((3.f * max_scale) / dvx), ceilf((3.f * max_scale) / dvy),
                                   ceilf((3.f * max_scale) / dvz));
    const float3 pv = voxel_position(p, v, dvx, dvy, dvz);
    if (pv.x + rad.x < 0 || pv.y + rad.y < 0 || pv.z + rad.z < 0 || pv.x - rad.x > (float)v.fnx ||
        pv.y - rad.y > (float)v.ny || pv.z - rad.z > (float)v.nz)
        return 0u;
    tile_cube(pv, rad, v.fgx, v.gy, v.gz, lo, hi);
    slab_clip(v, lo, hi);   // (a Gaussian without a tile in this slab is culled here: radii 0, no instances)
    const uint32_t n = (uint32_t)(hi.x - lo.x) * (uint32_t)(hi.y - lo.y) * (uint32_t)(hi.z - lo.z);
    if (n == 0) return 0u;

    radii_x[idx] = (int)rad.x;
    radii_y[idx] = (int)rad.y;
    radii_z[idx] = (int)rad.z;
    tiles_touched[idx] = n;
    // the tile cube, written here in Gaussian order (coalesced); the duplicate kernel, which walks the Gaussians in DEPTH order,
    // then needs one 16-byte gather per Gaussian instead of three radii + the record + the cube arithmetic (round 4: its
    // per-Gaussian part was 20 of its 38 us at 256^3) and fills in .x, the first instance
    if (cube)
        cube[idx] = make_uint4(0u, (uint32_t)lo.x | ((uint32_t)lo.y << 16), (uint32_t)lo.z | ((uint32_t)(hi.x - lo.x) << 16),
                               (uint32_t)(hi.y - lo.y));
    pv_out = pv;
    return n;
}

// Part 2, what the RENDER kernels read: the record {position, opacity, scaled inverse covariance, log2 opacity} and the culling
// extents of a visible Gaussian.
__device__ __forceinline__ void voxel_record_one(int idx, const float3 pv, const float *inv, const float *__restrict__ opacities,
                                                 float4 *__restrict__ rec, float4 *__restrict__ ext)
{
    const float inv_a = inv[0], inv_b = inv[1], inv_c = inv[2], inv_d = inv[3], inv_e = inv[4], inv_f = inv[5];
    const float op = opacities[idx];
    const float L = op > 0.0f ? log2f(op) : -INFINITY;
    // bounding box of {alpha >= 1e-6} (VOX/forward.cu:293): q = d^T C d <= 2 ln2 (L - log2(1e-6)), half-widths
    // sqrt(qmax * (C^-1)_kk), from the float inverse covariance the kernels evaluate, in double, padded; +inf (never
    // cull) unless C is safely positive definite, -inf (never live) when the opacity is below the cut-off.
    // Round 4: instead of the y / z half-widths of the whole box the record keeps the CROSS-SECTION of the cut-off ellipsoid
    // at a given x offset dx (what an x-slab of voxels sees): an ellipse centred at (y, z) = p_yz - (ky, kz) * dx whose own
    // bounding box has the half-widths (hyc, hzc) * sqrt(1 - (dx / hx)^2), hyc^2 = qmax * F / m00, hzc^2 = qmax * D / m00,
    // (ky, kz) = -[[D, E], [E, F]]^-1 (B, C).  The slab test built on it (slab_live, voxel_render.hip) drops the slabs in the
    // corners between the box and the ellipsoid: 21 % of the (instance, slab) pairs of a 256^3 query
    // (scripts/voxel_pair_fractions.py: slab_cond).
    float hx = INFINITY, hyc = INFINITY, hzc = INFINITY, ky = 0.f, kz = 0.f;
    {
        const double qmax = 2.0 * (double)LN2 * ((double)L - (double)LOG2_ALPHA_MIN_3D) + 1e-3;
        const double A = inv_a, B = inv_b, C = inv_c, D = inv_d, E = inv_e, F = inv_f;
        const double m00 = D * F - E * E, m11 = A * F - C * C, m22 = A * D - B * B;
        const double det3 = A * m00 - B * (B * F - C * E) + C * (B * E - C * D);
        if (!(qmax > 0.0)) {
            hx = hyc = hzc = -INFINITY;
        } else if (A > 0.0 && m22 > 0.0 && det3 > 0.0 && m00 > 0.0 && m11 > 0.0 && D > 0.0 && F > 0.0 &&
                   (A + D + F) * (m00 + m11 + m22) <= 1.0e4 * det3) {
            const double ex = sqrt(qmax * m00 / det3) * 1.004 + 0.05, ey = sqrt(qmax * F / m00) * 1.004 + 0.05,
                         ez = sqrt(qmax * D / m00) * 1.004 + 0.05;
            const double sy = -(F * B - E * C) / m00, sz = -(D * C - E * B) / m00;
            if (ex < 1.0e30 && ey < 1.0e30 && ez < 1.0e30 && fabs(sy) < 1.0e6 && fabs(sz) < 1.0e6) {
                hx = (float)ex; hyc = (float)ey; hzc = (float)ez; ky = (float)sy; kz = (float)sz;
            }
        }
    }
    rec[3 * idx] = make_float4(pv.x, pv.y, pv.z, op);
    rec[3 * idx + 1] = make_float4((-0.5f * LOG2E) * inv_a, (-LOG2E) * inv_b, (-LOG2E) * inv_c, (-0.5f * LOG2E) * inv_d);
    rec[3 * idx + 2] = make_float4((-LOG2E) * inv_e, (-0.5f * LOG2E) * inv_f, L, kz);
    ext[idx] = make_float4(hx, hyc, hzc, ky);
}

template <bool EARLY_CULL = false>
__device__ __forceinline__ void voxel_preprocess_one(
    int idx, const float *__restrict__ means3D, const float *__restrict__ scales, float scale_modifier,
    const float *__restrict__ rotations, const float *__restrict__ opacities, const float *__restrict__ cov3D_precomp,
    const VoxelGrid &v, int *__restrict__ radii_x, int *__restrict__ radii_y, int *__restrict__ radii_z,
    float4 *__restrict__ rec, uint32_t *__restrict__ depth_key,
    float *__restrict__ cov3Ds, uint32_t *__restrict__ tiles_touched, float4 *__restrict__ ext, const DepthReg &reg, uint32_t &key_out, uint2 &bt_out,
    uint4 *__restrict__ cube = nullptr)
{
    key_out = DEPTH_CULLED_KEY;
    float3 pv;
    float inv[6];
    int3 lo, hi;
    const uint32_t n = voxel_cull_one<EARLY_CULL>(idx, means3D, scales, scale_modifier, rotations, cov3D_precomp, v, radii_x, radii_y,
                                                  radii_z, depth_key, cov3Ds, tiles_touched, cube, pv, inv, lo, hi);
    if (n == 0u) return;
    key_out = __float_as_uint(means3D[3 * idx + 2]);   // visible: hinted depth order, the key goes straight into its bucket
    bt_out = depth_register_key(reg, key_out, n);
    voxel_record_one(idx, pv, inv, opacities, rec, ext);
}

__global__ void __launch_bounds__(256) voxel_preprocess_kernel(
    int P, const float *__restrict__ means3D, const float *__restrict__ scales, float scale_modifier,
    const float *__restrict__ rotations, const float *__restrict__ opacities, const float *__restrict__ cov3D_precomp,
    VoxelGrid v, int *__restrict__ radii_x, int *__restrict__ radii_y, int *__restrict__ radii_z,
    float4 *__restrict__ rec, uint32_t *__restrict__ depth_key,
    float *__restrict__ cov3Ds, uint32_t *__restrict__ tiles_touched, float4 *__restrict__ ext, DepthReg reg, uint4 *__restrict__ cube)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    uint32_t key = DEPTH_CULLED_KEY;
    uint2 bt = make_uint2(0u, 0u);
    if (idx < P)
        voxel_preprocess_one(idx, means3D, scales, scale_modifier, rotations, opacities, cov3D_precomp, v, radii_x, radii_y, radii_z,
                             rec, depth_key, cov3Ds, tiles_touched, ext, reg, key, bt, cube);
    depth_register_end(reg, (uint32_t)idx, key, bt);
}

// ---- stick-first chain (voxel_sticks.hip): part 1 of the preprocess for VS_PRODUCER Gaussians per workgroup + the workgroup's
// instance counts per LIST (stick of 2^sh consecutive tile ids) in an LDS histogram, stored as one row of H; the workgroup's
// totals: one plain word (the scatter kernel's row offsets) and ONE 64-bit atomic (the call's num_rendered).
__global__ void __launch_bounds__(VS_PRODUCER) voxel_cull_count_kernel(
    int P, uint32_t per_wg, uint32_t ni, const float *__restrict__ means3D, const float *__restrict__ scales, float scale_modifier,
    const float *__restrict__ rotations, const float *__restrict__ cov3D_precomp, VoxelGrid v, int *__restrict__ radii_x,
    int *__restrict__ radii_y, int *__restrict__ radii_z, uint32_t *__restrict__ depth_key, float *__restrict__ cov3Ds,
    uint32_t *__restrict__ tiles_touched, uint4 *__restrict__ cube, uint32_t sh, uint32_t stride, uint32_t *__restrict__ H,
    uint32_t *__restrict__ wgtot, VSCounters *__restrict__ ctr)
{
    extern __shared__ uint32_t s_hist[];   // [stride]
    __shared__ uint32_t s_w[2][VS_PRODUCER / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t i = tid; i < stride; i += VS_PRODUCER) s_hist[i] = 0u;
    __syncthreads();
    const uint32_t g0 = blockIdx.x * per_wg, g1 = min(g0 + per_wg, (uint32_t)P);
    uint32_t sum = 0u, vis = 0u;
    const uint32_t nsub = 1u << sh;
    auto count_row = [&](uint32_t t0, uint32_t rw) {   // the row's tiles, stick by stick
        const uint32_t t1 = t0 + rw - 1u;
        for (uint32_t l = t0 >> sh; l <= (t1 >> sh); ++l) {
            const uint32_t a = max(t0, l << sh), b = min(t1, (l << sh) + nsub - 1u);
            atomicAdd(&s_hist[l], b - a + 1u);
        }
    };
    for (uint32_t it = 0; it < ni; ++it) {
        const uint32_t idx = g0 + it * VS_PRODUCER + (uint32_t)tid;
        float3 pv;
        float inv[6];
        int3 lo = make_int3(0, 0, 0), hi = make_int3(0, 0, 0);
        uint32_t tt = 0u;
        if (idx < g1)
            tt = voxel_cull_one((int)idx, means3D, scales, scale_modifier, rotations, cov3D_precomp, v, radii_x, radii_y, radii_z, depth_key,
                                cov3Ds, tiles_touched, cube, pv, inv, lo, hi);
        sum += tt;
        vis += tt != 0u ? 1u : 0u;
        // (a Gaussian of many tiles: walked by the whole wave, a row per lane -- see the scatter kernel, voxel_sticks.hip)
        const bool big = tt > VS_BIG_GAUSSIAN;
        if (tt != 0u && !big)
            for (int z = lo.z; z < hi.z; ++z)
                for (int y = lo.y; y < hi.y; ++y)
                    count_row(((uint32_t)z * (uint32_t)v.gy + (uint32_t)y) * (uint32_t)v.gx + (uint32_t)lo.x, (uint32_t)(hi.x - lo.x));
        unsigned long long todo = __ballot(big);
        while (todo) {   // (wave-uniform)
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            const int ox = __shfl(lo.x, src), oy = __shfl(lo.y, src), oz = __shfl(lo.z, src);
            const uint32_t rw = (uint32_t)(__shfl(hi.x, src) - ox), rh = (uint32_t)(__shfl(hi.y, src) - oy),
                           rd = (uint32_t)(__shfl(hi.z, src) - oz);
            for (uint32_t r = (uint32_t)lane; r < rd * rh; r += 64u) {
                const uint32_t z = r / rh, y = r - z * rh;
                count_row((((uint32_t)oz + z) * (uint32_t)v.gy + (uint32_t)oy + y) * (uint32_t)v.gx + (uint32_t)ox, rw);
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        sum += (uint32_t)__shfl_xor(sum, d);
        vis += (uint32_t)__shfl_xor(vis, d);
    }
    if (lane == 0) { s_w[0][wave] = sum; s_w[1][wave] = vis; }
    __syncthreads();   // (also: the histogram is complete)
    if (tid == 0) {
        uint32_t S = 0, V = 0;
#pragma unroll
        for (int w = 0; w < (int)VS_PRODUCER / 64; ++w) { S += s_w[0][w]; V += s_w[1][w]; }
        wgtot[blockIdx.x] = S;
        atomicAdd(&ctr->total, ((unsigned long long)V << 40) | (unsigned long long)S);
    }
    uint32_t *__restrict__ row = H + (size_t)blockIdx.x * stride;
    for (uint32_t i = tid; i < stride; i += VS_PRODUCER) row[i] = s_hist[i];
}

// ... and part 2 for the visible ones (the inverse covariance is recomputed from the stored 3D covariance: the same operations on the
// same values, the same bits), in ONE launch with the chain's column scan -- two independent jobs, the scan on the first nscan
// workgroups (the layout of rs_scan_kernel, radix_sort.hip: a workgroup owns 32 consecutive lists, its 32 thread rows split the
// producer workgroups, every access is a full 128-byte row segment; its last workgroup posts the call's totals to the host).
constexpr int VSS_THREADS = (int)VS_PRODUCER, VSS_LISTS = 32, VSS_ROWS = VSS_THREADS / VSS_LISTS, VSS_BATCH = 16;
__device__ __forceinline__ void vs_scan_columns(uint32_t bx, uint32_t nblocks, uint32_t *__restrict__ H, uint32_t rows, uint32_t stride,
                                                uint32_t *__restrict__ totals, VSCounters *__restrict__ ctr, uint32_t *__restrict__ words,
                                                uint32_t *__restrict__ mailbox, uint32_t seq)
{
    __shared__ uint32_t part[VSS_ROWS][VSS_LISTS];
    const uint32_t dl = threadIdx.x % VSS_LISTS, row = threadIdx.x / VSS_LISTS;
    const uint32_t d = bx * VSS_LISTS + dl;   // < stride (a multiple of 32)
    const uint32_t per = (rows + VSS_ROWS - 1) / VSS_ROWS;
    const uint32_t t0 = min(rows, row * per), t1 = min(rows, t0 + per);
    uint32_t sum = 0;
    for (uint32_t t = t0; t < t1; t += VSS_BATCH) {
        uint32_t v[VSS_BATCH];
#pragma unroll
        for (int u = 0; u < VSS_BATCH; ++u) v[u] = H[(size_t)min(t + (uint32_t)u, t1 - 1u) * stride + d];
#pragma unroll
        for (int u = 0; u < VSS_BATCH; ++u) sum += (t + (uint32_t)u < t1) ? v[u] : 0u;
    }
    part[row][dl] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (int r = 0; r < VSS_ROWS; ++r) {
        const uint32_t v = part[r][dl];
        if ((uint32_t)r < row) run += v;
        total += v;
    }
    for (uint32_t t = t0; t < t1; t += VSS_BATCH) {
        uint32_t v[VSS_BATCH];
#pragma unroll
        for (int u = 0; u < VSS_BATCH; ++u) v[u] = H[(size_t)min(t + (uint32_t)u, t1 - 1u) * stride + d];
#pragma unroll
        for (int u = 0; u < VSS_BATCH; ++u)
            if (t + (uint32_t)u < t1) {
                H[(size_t)(t + (uint32_t)u) * stride + d] = run;
                run += v[u];
            }
    }
    if (row == 0) totals[d] = total;
    // the longest list of the call; the last workgroup to get here tells the host
    if (threadIdx.x < 64) {
        uint32_t m = row == 0 ? total : 0u;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) m = max(m, (uint32_t)__shfl_xor(m, s));
        if (threadIdx.x == 0) {
            atomicMax(&ctr->maxlist, m);
            __threadfence();
            const uint32_t done = atomicAdd(&ctr->scan_done, 1u);
            if (done == nblocks - 1u) {
                __threadfence();
                const uint32_t ml = atomicMax(&ctr->maxlist, 0u);
                const unsigned long long tot = atomicAdd(&ctr->total, 0ull);
                const unsigned long long r40 = tot & ((1ull << 40) - 1ull);
                const uint32_t R = r40 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)r40;
                // DW_NVIS = 0: `order` will hold all P ids (the geometry backward walks all of it)
                words[DW_TOTAL] = R; words[DW_OVERFLOW] = 0u; words[DW_USER] = VOX_STICKS_MARK; words[DW_PMAX] = ml;
                words[DW_PNMAX] = 0u; words[DW_NMAX] = 0u; words[DW_NNMAX] = 0u; words[DW_NVIS] = 0u;
                mailbox[DW_TOTAL] = R; mailbox[DW_OVERFLOW] = 0u; mailbox[DW_USER] = VOX_STICKS_MARK; mailbox[DW_PMAX] = ml;
                mailbox[DW_PNMAX] = 0u; mailbox[DW_NMAX] = 0u; mailbox[DW_NNMAX] = 0u; mailbox[DW_NVIS] = (uint32_t)(tot >> 40);
                __hip_atomic_store(&mailbox[15], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                // every other workgroup is through with the counters: ready for the thread's next call on this stream
                __hip_atomic_store(&ctr->total, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ctr->maxlist, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ctr->scan_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

__global__ void __launch_bounds__(VS_PRODUCER) voxel_scan_records_kernel(
    uint32_t nscan, uint32_t *__restrict__ H, uint32_t rows, uint32_t stride, uint32_t *__restrict__ totals, VSCounters *__restrict__ ctr,
    uint32_t *__restrict__ words, uint32_t *__restrict__ mailbox, uint32_t seq, int P, uint32_t per_wg, uint32_t ni,
    const float *__restrict__ means3D, const float *__restrict__ opacities, const float *__restrict__ cov3Ds,
    const uint32_t *__restrict__ tiles_touched, VoxelGrid v, float4 *__restrict__ rec, float4 *__restrict__ ext)
{
    if (blockIdx.x < nscan) {
        vs_scan_columns(blockIdx.x, nscan, H, rows, stride, totals, ctr, words, mailbox, seq);
        return;
    }
    const uint32_t wg = blockIdx.x - nscan;   // the producers' mapping (vs_grid): one workgroup per CU
    const uint32_t g0 = wg * per_wg, g1 = min(g0 + per_wg, (uint32_t)P);
    const float dvx = v.sx / (float)v.fnx, dvy = v.sy / (float)v.ny, dvz = v.sz / (float)v.nz;
    for (uint32_t it = 0; it < ni; ++it) {
        const uint32_t idx = g0 + it * VS_PRODUCER + threadIdx.x;
        if (idx >= g1 || tiles_touched[idx] == 0u) continue;
        const float3 p = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
        float cov3D[6], inv[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) cov3D[k] = cov3Ds[6 * idx + k];
        if (!voxel_inverse(cov3D, dvx, dvy, dvz, inv)) continue;   // (cannot happen: part 1 gave it tiles)
        voxel_record_one((int)idx, voxel_position(p, v, dvx, dvy, dvz), inv, opacities, rec, ext);
    }
}

int launch_voxel_cull_count(const VoxelGeom &g, const VoxelGrid &v, int P, const VSGrid &grid, const float *means3D, const float *scales,
                            float scale_modifier, const float *rotations, const float *cov3D_precomp, int *radii_x, int *radii_y,
                            int *radii_z, uint32_t shift, uint32_t stride, uint32_t *H, uint32_t *wgtot, VSCounters *ctr, hipStream_t s)
{
    voxel_cull_count_kernel<<<dim3(grid.wgs), dim3(VS_PRODUCER), stride * sizeof(uint32_t), s>>>(
        P, grid.per_wg, grid.ni, means3D, scales, scale_modifier, rotations, cov3D_precomp, v, radii_x, radii_y, radii_z, g.depth_key,
        g.cov3D, g.tiles_touched, g.cube, shift, stride, H, wgtot, ctr);
    return 0;
}

int launch_voxel_scan_records(const VoxelGeom &g, const VoxelGrid &v, int P, const VSGrid &grid, const float *means3D,
                              const float *opacities, const float *cov3D_precomp, uint32_t *H, uint32_t rows, uint32_t stride,
                              uint32_t *totals, VSCounters *ctr, uint32_t *mailbox, uint32_t seq, hipStream_t s)
{
    const uint32_t nscan = stride / VSS_LISTS;
    voxel_scan_records_kernel<<<dim3(nscan + grid.wgs), dim3(VS_PRODUCER), 0, s>>>(
        nscan, H, rows, stride, totals, ctr, g.host_words, mailbox, seq, P, grid.per_wg, grid.ni, means3D, opacities,
        cov3D_precomp ? cov3D_precomp : g.cov3D, g.tiles_touched, v, g.rec, g.ext);
    return 0;
}

// ---- small grids (the training loop's 32^3 TV patch: 64 tiles; train.py:128-142).  Only ~2 % of the Gaussians reach such a
// patch, and the general pipeline spends its time on per-Gaussian bookkeeping over ALL of them (bucket counters, dual scan,
// place, rank, emission, tile sort: nine latency-bound launches).  Here the preprocess itself hands every
// SURVIVOR (a) its run of rows in the backward's moment scratch and (b) a slot in a compact survivor list -- one 64-bit atomic
// per 1024-thread workgroup {workgroups done : 12 | survivors : 20 | rows : 32} (same-address atomics retire at ~90 per
// microsecond device-wide: one per wave would cost as much as the whole kernel) -- and the LAST workgroup posts the totals to
// the host mailbox.  voxel_small.hip then builds every tile's depth-sorted list straight from the survivor list.  Nothing of
// the reference's contract changes: radii, tiles_touched, num_rendered, point_list and ranges are bit-identical; the order in
// which Gaussians get their scratch rows is ours (any disjoint assignment serves the backward).
__global__ void __launch_bounds__(1024) voxel_preprocess_small_kernel(
    int P, const float *__restrict__ means3D, const float *__restrict__ scales, float scale_modifier,
    const float *__restrict__ rotations, const float *__restrict__ opacities, const float *__restrict__ cov3D_precomp,
    VoxelGrid v, int *__restrict__ radii_x, int *__restrict__ radii_y, int *__restrict__ radii_z,
    float4 *__restrict__ rec, uint32_t *__restrict__ depth_key, float *__restrict__ cov3Ds,
    uint32_t *__restrict__ tiles_touched, float4 *__restrict__ ext, uint32_t *__restrict__ first, uint4 *__restrict__ cube,
    uint32_t *__restrict__ order, uint4 *__restrict__ surv, unsigned long long *counter /* persistent, zero between calls */, uint32_t *__restrict__ words,
    uint32_t *__restrict__ mailbox, uint32_t seq)
{
    const int idx = blockIdx.x * 1024 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t key = DEPTH_CULLED_KEY;
    uint2 bt = make_uint2(0u, 0u);
    const DepthReg noreg{};
    if (idx < P)
        voxel_preprocess_one<true>(idx, means3D, scales, scale_modifier, rotations, opacities, cov3D_precomp, v, radii_x, radii_y,
                                   radii_z, rec, depth_key, cov3Ds, tiles_touched, ext, noreg, key, bt);
    const bool vis = key != DEPTH_CULLED_KEY;
    const uint32_t n = vis ? tiles_touched[idx] : 0u;
    // exclusive prefix of (rows, survivors) inside the workgroup
    uint32_t in = n, iv = vis ? 1u : 0u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t un = __shfl_up(in, d), uv = __shfl_up(iv, d);
        if (lane >= d) { in += un; iv += uv; }
    }
    __shared__ uint32_t wn[16], wv[16];
    __shared__ unsigned long long s_old;
    if (lane == 63) { wn[wave] = in; wv[wave] = iv; }
    __syncthreads();
    uint32_t bn = 0, bv = 0, tn = 0, tv = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        if (w < wave) { bn += wn[w]; bv += wv[w]; }
        tn += wn[w]; tv += wv[w];
    }
    if (threadIdx.x == 0) s_old = atomicAdd(counter, (1ull << 52) | ((unsigned long long)tv << 32) | (unsigned long long)tn);
    __syncthreads();
    const unsigned long long old = s_old;
    const uint32_t row0 = (uint32_t)old, sv0 = (uint32_t)(old >> 32) & 0xFFFFFu, done = (uint32_t)(old >> 52);
    if (vis) {
        const float4 r0 = rec[3 * idx];
        int3 lo, hi;
        tile_cube(make_float3(r0.x, r0.y, r0.z), make_float3((float)radii_x[idx], (float)radii_y[idx], (float)radii_z[idx]),
                  v.gx, v.gy, v.gz, lo, hi);
        const uint32_t f = row0 + bn + in - n;
        first[idx] = f;
        cube[idx] = make_uint4(f, (uint32_t)lo.x | ((uint32_t)lo.y << 16), (uint32_t)lo.z | ((uint32_t)(hi.x - lo.x) << 16),
                               (uint32_t)(hi.y - lo.y));
        order[sv0 + bv + iv - 1u] = (uint32_t)idx;   // the compact list of visible ids (the geometry backward walks it)
        surv[sv0 + bv + iv - 1u] = make_uint4((uint32_t)idx, key,
                                               (uint32_t)lo.x | ((uint32_t)lo.y << 4) | ((uint32_t)lo.z << 8) | ((uint32_t)hi.x << 12) |
                                                   ((uint32_t)hi.y << 16) | ((uint32_t)hi.z << 20), f);
    }
    if (done == gridDim.x - 1u && threadIdx.x == 0) {   // the last workgroup to arrive: totals -> state + host, counter back to 0
        const uint32_t R = row0 + tn, nsurv = sv0 + tv;
        *counter = 0ull;
        words[DW_TOTAL] = R; words[DW_OVERFLOW] = 0u; words[DW_USER] = VOX_SMALL_MARK; words[DW_NVIS] = nsurv;
        words[DW_PMAX] = 0u; words[DW_PNMAX] = 0u; words[DW_NMAX] = 0u; words[DW_NNMAX] = 0u;
        mailbox[DW_TOTAL] = R; mailbox[DW_OVERFLOW] = 0u; mailbox[DW_USER] = VOX_SMALL_MARK; mailbox[DW_NVIS] = nsurv;
        mailbox[DW_PMAX] = 0u; mailbox[DW_PNMAX] = 0u; mailbox[DW_NMAX] = 0u; mailbox[DW_NNMAX] = 0u;
        __hip_atomic_store(&mailbox[15], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int launch_voxel_preprocess_small(const VoxelGeom &g, const VoxelGrid &v, int P, const float *means3D, const float *scales,
                                  float scale_modifier, const float *rotations, const float *opacities,
                                  const float *cov3D_precomp, int *radii_x, int *radii_y, int *radii_z, uint4 *surv,
                                  unsigned long long *counter, uint32_t *mailbox, uint32_t seq, hipStream_t s)
{
    voxel_preprocess_small_kernel<<<dim3((P + 1023) / 1024), dim3(1024), 0, s>>>(
        P, means3D, scales, scale_modifier, rotations, opacities, cov3D_precomp, v, radii_x, radii_y, radii_z, g.rec, g.depth_key,
        g.cov3D, g.tiles_touched, g.ext, g.first, g.cube, g.order, surv, counter, g.host_words, mailbox, seq);
    return 0;
}

// Instance emission (duplicateWithKeys, VOX/voxelizer_impl.cu:54-101) in DEPTH order: sorted position j ->
// Gaussian order[j] -> its tiles z-major / y / x-minor.  Only the tile id is the sort key (see binning.hip).
// One wave serves 64 consecutive sorted positions and walks their contiguous output span with coalesced
// stores (see raster_geom.hip).
__global__ void __launch_bounds__(256) voxel_duplicate_kernel(
    int P, const float4 *__restrict__ rec, const uint32_t *__restrict__ order, const uint32_t *__restrict__ offsets,
    const int *__restrict__ radii_x, const int *__restrict__ radii_y, const int *__restrict__ radii_z, int gx, int gy,
    int gz, uint32_t *__restrict__ first, uint4 *__restrict__ cube, uint32_t *__restrict__ tiles, uint32_t *__restrict__ vals,
    const uint32_t *__restrict__ nvis, uint2 *__restrict__ zero_ranges, uint32_t zero_T)
{
    // side job: the tile ranges start out empty (tile_ranges_kernel only writes the tiles that hold instances) -- a separate
    // fill launch costs ~5 us
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < zero_T; i += gridDim.x * 256u) zero_ranges[i] = make_uint2(0u, 0u);
    if (nvis) P = min(P, (int)*nvis);   // hinted depth order: only the visible prefix of order / offsets is written
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave_first = j - lane;
    if (wave_first >= P) return;
    uint32_t id = 0;
    uint32_t excl, incl;
    if (j < P) {
        id = order[j];
        incl = offsets[j];
        excl = j == 0 ? 0u : offsets[j - 1];
    } else {
        incl = excl = offsets[P - 1];
    }
    // a Gaussian emits instances <=> it is visible; its tile cube was left by the preprocess (see there)
    int3 lo = make_int3(0, 0, 0);
    int rw = 0, rh = 0;
    if (incl != excl) {
        const uint4 c = cube[id];
        lo = make_int3((int)(c.y & 0xFFFFu), (int)(c.y >> 16), (int)(c.z & 0xFFFFu));
        rw = (int)(c.z >> 16);
        rh = (int)c.w;
        first[id] = excl;
        reinterpret_cast<uint32_t *>(cube + id)[0] = excl;
    }
    const uint32_t wbeg = __shfl(excl, 0);
    const int last_lane = min(63, P - 1 - wave_first);
    const uint32_t wend = __shfl(incl, last_lane);
    for (uint32_t base = wbeg; base < wend; base += 64) {
        const uint32_t k = base + lane;
        // (Measured and left out, round 4: the owner as a count -- ballots for the window's first slot, the few run starts inside
        // the window read with v_readlane and counted per slot -- instead of this six-step search: 28.7 -> 31.0 us.)
        int own = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {
            const int probe = own + step;
            const uint32_t e = __shfl(excl, probe & 63);
            if (probe <= last_lane && e <= k) own = probe;
        }
        const uint32_t o_excl = __shfl(excl, own);
        const int ox = __shfl(lo.x, own), oy = __shfl(lo.y, own), oz = __shfl(lo.z, own);
        const int orw = __shfl(rw, own), orh = __shfl(rh, own);
        const uint32_t o_id = __shfl(id, own);
        if (k < wend) {
            const uint32_t local = k - o_excl;
            const uint32_t xy = (uint32_t)orw * (uint32_t)orh;
            // local / xy and rem / orw through the float reciprocal: floor((n + 0.5) * rcp(d)) is exact while the product's
            // rounding error (~1.2e-7 n / d) stays below the 0.5 / d that separates (n + 0.5) / d from an integer, i.e. for
            // n < 4e6 -- a Gaussian's cube holds a few hundred tiles (the u32 division is ~25 instructions, twice per instance)
            uint32_t qz, qy;
            if (local < (1u << 20)) {
                qz = (uint32_t)(((float)local + 0.5f) * __builtin_amdgcn_rcpf((float)xy));
                const uint32_t rem0 = local - qz * xy;
                qy = (uint32_t)(((float)rem0 + 0.5f) * __builtin_amdgcn_rcpf((float)orw));
            } else {
                qz = local / xy;
                qy = (local - qz * xy) / (uint32_t)orw;
            }
            const uint32_t rem = local - qz * xy;
            const int z = oz + (int)qz;
            const int y = oy + (int)qy;
            const int x = ox + (int)(rem - qy * (uint32_t)orw);
            tiles[k] = (uint32_t)(z * gy * gx + y * gx + x);
            vals[k] = o_id;
        }
    }
}

// zero the K-float rows of a wave's 64 Gaussians whose bit is set in `rows`, with unit-stride stores over the wave's span
template <int K>
__device__ __forceinline__ void zero_culled_rows(float *__restrict__ a, size_t row0, int lane, unsigned long long rows)
{
    if (a == nullptr) return;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int e = j * 64 + lane;
        if ((rows >> (e / K)) & 1ull) a[row0 * (size_t)K + (size_t)e] = 0.f;
    }
}

// Fused geometry backward, one pass per Gaussian:
//   1. reduce the per-instance moment rows of the render backward (contiguous run in the emission list),
//      fixed order -> deterministic, atomic-free (reference: 10 float atomicAdd per pair, VOX/backward.cu:359-370);
//   2. moments -> dL/dmean3D_norm (x dVoxel, quirk Q4), dL/dconic3D, dL/dopacity;
//   3. computeCov3DCUDA (VOX/backward.cu:86-177) + preprocessCUDA backward (VOX/backward.cu:180-213).
// Two launches: the zero rows of the culled Gaussians (every lane, unit-stride stores), then the real work over the COMPACT list
// of visible ids -- `order[0 .. nvis)`, which every forward path leaves behind (depth order: hinted prefix / full permutation with
// the culled ids last / the small-grid survivor list).  On the training loop's TV patch 2 % of the Gaussians are visible, spread
// thinly over the waves: with one lane per Gaussian nearly every wave ran the whole chain below for one or two lanes.
__global__ void __launch_bounds__(256) voxel_geom_backward_zero_kernel(
    int P, const int *__restrict__ radii_x, const int *__restrict__ radii_y, const int *__restrict__ radii_z,
    float *__restrict__ dL_dconic3D, float *__restrict__ dL_dmean3D_norm, float *__restrict__ dL_dopacity,
    float *__restrict__ dL_dmeans, float *__restrict__ dL_dcov, float *__restrict__ dL_dscale, float *__restrict__ dL_drot,
    int visible_get_zero_scale_rot)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    // culled Gaussians: all-zero gradient rows (the reference relies on zero-filled tensors, SUB/voxelize_points.cu:130-136).
    // The WAVE writes them: its 64 rows are one contiguous span of every output array, walked with unit-stride stores, each
    // element zeroed iff its row's lane is culled
    const bool in_range = idx < P;
    const bool culled = in_range && (!(radii_x[idx] > 0) || !(radii_y[idx] > 0) || !(radii_z[idx] > 0));
    const unsigned long long cmask = __ballot(culled);
    const unsigned long long all = __ballot(in_range);
    const int lane = threadIdx.x & 63;
    const size_t row0 = (size_t)(idx - lane);
    if (cmask) {   // wave-uniform
        zero_culled_rows<3>(dL_dmean3D_norm, row0, lane, cmask);
        zero_culled_rows<3>(dL_dmeans, row0, lane, cmask);
        zero_culled_rows<6>(dL_dconic3D, row0, lane, cmask);
        zero_culled_rows<6>(dL_dcov, row0, lane, cmask);
        zero_culled_rows<1>(dL_dopacity, row0, lane, cmask);
    }
    // scale / rotation gradients: zero for the culled rows, and for ALL rows when the covariance was given (cov3D_precomp)
    const unsigned long long zmask = visible_get_zero_scale_rot ? all : cmask;
    if (zmask) {
        zero_culled_rows<3>(dL_dscale, row0, lane, zmask);
        zero_culled_rows<4>(dL_drot, row0, lane, zmask);
    }
}

// Small grids (nearly everything is culled): ALL rows are zeroed with plain wide stores -- the compact kernel that follows
// overwrites the visible ones (ordered by the kernel boundary).  The predicated 4-byte version above moves 31 MB at ~2 TB/s.
// (Round 4: on a patch with instances the zero-fill rides along with the render backward instead -- extra workgroups of that
// launch, voxel_render.hip -- so this kernel only serves patches nothing reaches.)
__global__ void __launch_bounds__(256) voxel_zero_all_rows_kernel(ZeroArrays z)
{
    const size_t t = (size_t)blockIdx.x * 256u + threadIdx.x, nt = (size_t)gridDim.x * 256u;
#pragma unroll
    for (int a = 0; a < 7; ++a) {
        float *__restrict__ p = z.p[a];
        if (p == nullptr) continue;
        const size_t n = z.n[a];
        const size_t head = min(n, (size_t)((16u - ((uintptr_t)p & 15u)) & 15u) / 4u);   // floats before the first 16-byte boundary
        float4 *__restrict__ p4 = reinterpret_cast<float4 *>(p + head);
        const size_t n4 = (n - head) / 4;
        for (size_t i = t; i < n4; i += nt) p4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < head) p[t] = 0.f;
        const size_t tail0 = head + 4 * n4;
        if (t < n - tail0) p[tail0 + t] = 0.f;
    }
}

__global__ void __launch_bounds__(256) voxel_geom_backward_kernel(
    int P, const uint32_t *__restrict__ order, const uint32_t *__restrict__ words,
    const int *__restrict__ radii_x, const int *__restrict__ radii_y, const int *__restrict__ radii_z,
    const float *__restrict__ cov3Ds, const float *__restrict__ scales, const float *__restrict__ rotations,
    float scale_modifier, VoxelGrid v, const float4 *__restrict__ rec, const uint32_t *__restrict__ first_inst,
    const uint32_t *__restrict__ tiles_touched, const float4 *__restrict__ part,
    float *__restrict__ dL_dconic3D,
    float *__restrict__ dL_dmean3D_norm, float *__restrict__ dL_dopacity, float *__restrict__ dL_dmeans,
    float *__restrict__ dL_dcov, float *__restrict__ dL_dscale, float *__restrict__ dL_drot)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    // number of list entries that can be visible: the hinted depth order / the small-grid path record it (DW_NVIS); the
    // un-hinted orders are full permutations with the culled ids at the tail (DW_NVIS == 0: walk all of it)
    const uint32_t nvis = words[DW_NVIS];
    const int count = nvis ? (int)min(nvis, (uint32_t)P) : P;
    if (j >= count) return;
    const int idx = (int)order[j];
    if ((uint32_t)idx >= (uint32_t)P) return;   // nothing visible at all: the list was never written (every row is a zero row)
    if (!(radii_x[idx] > 0) || !(radii_y[idx] > 0) || !(radii_z[idx] > 0)) return;
    const float dvx = v.sx / (float)v.fnx, dvy = v.sy / (float)v.ny, dvz = v.sz / (float)v.nz;

    // ---- 1. moments: S0, (Sx,Sy,Sz), (Sxx,Sxy,Sxz,Syy,Syz,Szz)
    const uint32_t first = first_inst[idx], ninst = tiles_touched[idx];
    float S[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) S[k] = 0.f;
    // the render backward stores each instance's row at its emission index: this Gaussian's rows are contiguous.  Four rows
    // per trip, all twelve loads in flight before the first add (the adds keep the row order: bit-reproducible); a wave's
    // trip count is that of its widest Gaussian (up to 64 tiles on the TV patch)
    for (uint32_t j = 0; j < ninst; j += 4) {
        float4 m0[4], m1[4], m2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t row = (size_t)first + min(j + (uint32_t)i, ninst - 1u);   // clamped: branch-free loads
            m0[i] = part[3 * row];
            m1[i] = part[3 * row + 1];
            m2[i] = part[3 * row + 2];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (j + (uint32_t)i < ninst) {
                S[0] += m0[i].x; S[1] += m0[i].y; S[2] += m0[i].z; S[3] += m0[i].w;
                S[4] += m1[i].x; S[5] += m1[i].y; S[6] += m1[i].z; S[7] += m1[i].w;
                S[8] += m2[i].x; S[9] += m2[i].y;
            }
    }
    // ---- 2. the reference's accumulated sums (VOX/backward.cu:345-370), inverse covariance un-scaled
    const float4 r0 = rec[3 * idx], r1 = rec[3 * idx + 1], r2 = rec[3 * idx + 2];
    const float opa = r0.w;
    const float ia = r1.x * (-2.0f * LN2), ib = r1.y * (-LN2), ic = r1.z * (-LN2);
    const float id_ = r1.w * (-2.0f * LN2), ie = r2.x * (-LN2), if_ = r2.y * (-2.0f * LN2);
    const float gmx = opa * dvx * (-ia * S[1] - ib * S[2] - ic * S[3]);
    const float gmy = opa * dvy * (-id_ * S[2] - ib * S[1] - ie * S[3]);
    const float gmz = opa * dvz * (-if_ * S[3] - ic * S[1] - ie * S[2]);
    const float ga = -0.5f * opa * S[4], gb = -opa * S[5], gc = -opa * S[6];
    const float gd = -0.5f * opa * S[7], ge = -opa * S[8], gf = -0.5f * opa * S[9];
    dL_dmean3D_norm[3 * idx + 0] = gmx;
    dL_dmean3D_norm[3 * idx + 1] = gmy;
    dL_dmean3D_norm[3 * idx + 2] = gmz;
    dL_dconic3D[6 * idx + 0] = ga; dL_dconic3D[6 * idx + 1] = gb; dL_dconic3D[6 * idx + 2] = gc;
    dL_dconic3D[6 * idx + 3] = gd; dL_dconic3D[6 * idx + 4] = ge; dL_dconic3D[6 * idx + 5] = gf;
    dL_dopacity[idx] = S[0];

    // ---- 3. geometry chain
    float cov3D[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) cov3D[k] = cov3Ds[6 * idx + k];
    M3 M;
    float h[6];
    voxel_cov(cov3D, dvx, dvy, dvz, M, h);
    const float hata = h[0], hatb = h[1], hatc = h[2], hatd = h[3], hate = h[4], hatf = h[5];
    const float denom = hata * hatd * hatf + 2 * hatb * hatc * hate - hata * hate * hate - hatf * hatb * hatb - hatd * hatc * hatc;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float o[6] = { 0, 0, 0, 0, 0, 0 };
    if (denom2inv != 0) {
        const float denom_da = hatd * hatf - hate * hate;
        const float denom_db = 2 * hatc * hate - 2 * hatf * hatb;
        const float denom_dc = 2 * hatb * hate - 2 * hatd * hatc;
        const float denom_dd = hata * hatf - hatc * hatc;
        const float denom_de = 2 * hatb * hatc - 2 * hata * hate;
        const float denom_df = hata * hatd - hatb * hatb;
        const float ce_bf = hatc * hate - hatb * hatf;
        const float be_cd = hatb * hate - hatc * hatd;
        const float bc_ae = hatb * hatc - hata * hate;
        const float da = denom2inv * (-denom_da*denom_da*ga - ce_bf*denom_da*gb - be_cd*denom_da*gc + (hatf*denom-denom_dd*denom_da)*gd + (-hate*denom-bc_ae*denom_da)*ge + (hatd*denom-denom_df*denom_da)*gf);
        const float db = denom2inv * (-denom_da*denom_db*ga + (-hatf*denom-ce_bf*denom_db)*gb + (hate*denom-be_cd*denom_db)*gc - denom_dd*denom_db*gd + (hatc*denom-bc_ae*denom_db)*ge + (-2*hatb*denom-denom_df*denom_db)*gf);
        const float dc = denom2inv * (-denom_da*denom_dc*ga + (hate*denom-ce_bf*denom_dc)*gb + (-hatd*denom-be_cd*denom_dc)*gc + (-2*hatc*denom-denom_dd*denom_dc)*gd + (hatb*denom-bc_ae*denom_dc)*ge - denom_df*denom_dc*gf);
        const float dd = denom2inv * ((hatf*denom-denom_da*denom_dd)*ga - ce_bf*denom_dd*gb +(-hatc*denom-be_cd*denom_dd)*gc - denom_dd*denom_dd*gd - bc_ae*denom_dd*ge + (hata*denom-denom_df*denom_dd)*gf);
        const float de = denom2inv * ((-2*hate*denom-denom_da*denom_de)*ga + (hatc*denom-ce_bf*denom_de)*gb + (hatb*denom-be_cd*denom_de)*gc - denom_dd*denom_de*gd + (-hata*denom-bc_ae*denom_de)*ge + -denom_df*denom_de*gf);
        const float df = denom2inv * ((hatd*denom-denom_da*denom_df)*ga + (-hatb*denom-ce_bf*denom_df)*gb - be_cd*denom_df*gc + (hata*denom-denom_dd*denom_df)*gd - bc_ae*denom_df*ge - denom_df*denom_df*gf);
        dcov_from_dhat(M, da, db, dc, dd, de, df, o);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) dL_dcov[6 * idx + k] = o[k];
    dL_dmeans[3 * idx + 0] = gmx;   // zero-init + "+= dL_dmean3D_norm" in the reference
    dL_dmeans[3 * idx + 1] = gmy;
    dL_dmeans[3 * idx + 2] = gmz;
    if (scales != nullptr && rotations != nullptr) {
        float ds[3];
        float4 dq;
        cov3d_backward(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2], scale_modifier,
                       reinterpret_cast<const float4 *>(rotations)[idx], o, ds, &dq);
        dL_dscale[3 * idx + 0] = ds[0];
        dL_dscale[3 * idx + 1] = ds[1];
        dL_dscale[3 * idx + 2] = ds[2];
        reinterpret_cast<float4 *>(dL_drot)[idx] = dq;
    }   // else (cov3D_precomp): the zero kernel has written the scale / rotation rows
}

int launch_voxel_preprocess(const VoxelGeom &g, const VoxelGrid &v, int P, const float *means3D, const float *scales,
                            float scale_modifier, const float *rotations, const float *opacities,
                            const float *cov3D_precomp, int *radii_x, int *radii_y, int *radii_z, const DepthReg &reg,
                            bool store_cov3D, hipStream_t s)
{
    voxel_preprocess_kernel<<<dim3((P + 255) / 256), dim3(256), 0, s>>>(P, means3D, scales, scale_modifier, rotations,
                                                                        opacities, cov3D_precomp, v, radii_x, radii_y,
                                                                        radii_z, g.rec, g.depth_key, store_cov3D ? g.cov3D : nullptr,
                                                                        g.tiles_touched, g.ext, reg, g.cube);
    return 0;
}

int launch_voxel_duplicate(const VoxelGeom &g, const VoxelBinning &b, const VoxelGrid &v, int P, const int *radii_x,
                           const int *radii_y, const int *radii_z, const uint32_t *nvis, hipStream_t s, uint2 *zero_ranges,
                           size_t zero_T)
{
    voxel_duplicate_kernel<<<dim3((P + 255) / 256), dim3(256), 0, s>>>(P, g.rec, g.order, g.offsets, radii_x, radii_y,
                                                                       radii_z, v.gx, v.gy, v.gz, g.first, g.cube,
                                                                       b.tiles_unsorted, b.vals_unsorted, nvis, zero_ranges,
                                                                       zero_ranges ? (uint32_t)zero_T : 0u);
    return 0;
}

int launch_voxel_geom_backward(const VoxelGeom &g, const VoxelGrid &v, int P, const int *radii_x, const int *radii_y,
                               const int *radii_z, const float *cov3D, const float *scales, const float *rotations,
                               float scale_modifier, const float *part, float *dL_dconic3D,
                               float *dL_dmean3D_norm,
                               float *dL_dopacity, float *dL_dmean3D, float *dL_dcov3D, float *dL_dscale, float *dL_drot,
                               hipStream_t s, bool rows_already_zero)
{
    if (rows_already_zero) {
        // the render backward's launch zeroed every row (patches, see launch_voxel_render_backward)
    } else if ((size_t)v.gx * v.gy * v.gz <= VOX_SMALL_MAX_TILES) {   // a patch: nearly every row is a zero row
        const size_t Pz = (size_t)P;
        const ZeroArrays z{{dL_dmean3D_norm, dL_dmean3D, dL_dconic3D, dL_dcov3D, dL_dopacity, dL_dscale, dL_drot},
                           {3 * Pz, 3 * Pz, 6 * Pz, 6 * Pz, Pz, 3 * Pz, 4 * Pz}};
        voxel_zero_all_rows_kernel<<<dim3(1024), dim3(256), 0, s>>>(z);
    } else {
        voxel_geom_backward_zero_kernel<<<dim3((P + 255) / 256), dim3(256), 0, s>>>(
            P, radii_x, radii_y, radii_z, dL_dconic3D, dL_dmean3D_norm, dL_dopacity, dL_dmean3D, dL_dcov3D, dL_dscale, dL_drot,
            (scales == nullptr || rotations == nullptr) ? 1 : 0);
    }
    voxel_geom_backward_kernel<<<dim3((P + 255) / 256), dim3(256), 0, s>>>(
        P, g.order, g.host_words, radii_x, radii_y, radii_z, cov3D, scales, rotations, scale_modifier, v, g.rec, g.first,
        g.tiles_touched, reinterpret_cast<const float4 *>(part), dL_dconic3D, dL_dmean3D_norm, dL_dopacity, dL_dmean3D, dL_dcov3D,
        dL_dscale, dL_drot);
    return 0;
}

}  // namespace r2
