// raster_render.hip -- per-tile additive line-integral render of the X-ray rasterizer and its backward.
//
// Reference: renderCUDA forward RAS/forward.cu:294-395, renderCUDA backward RAS/backward.cu:447-575.
// These kernels are VALU/exp-bound (256 pixel-Gaussian pairs per 32 bytes gathered; ~15 VALU issue slots
// per pair, v_exp_f32 alone costs ~7 of them), so this file is compiled with FMA contraction ON and
// without SLP packing; results are tolerance-checked, not bit-checked.
//
// Forward : a tile's depth-sorted list is cut into work items of FWD_CHUNK instances.  One workgroup =
//           one work item = 4 waves of 16x4 pixels; records are staged through LDS in 256-record batches
//           (32-byte packed records, two b128 gathers each) and read back as wave-uniform broadcasts.
//           Each work item writes 256 partial pixel sums; a second tiny kernel adds a tile's partials IN
//           LIST ORDER, so the image is deterministic.  Cutting the lists is what balances the machine:
//           list lengths span 0..9000 on the benchmark scene (median 50), and a workgroup per tile left
//           most CUs idle behind a few dense tiles.
// Backward: the loop nest is inverted.  One LANE owns one (tile, Gaussian) instance of the sorted list and
//           walks the 256 pixels of its tile; dL/dpix of the tile is staged once per wave in LDS and read
//           as b128 broadcasts.  The 7 gradient sums of the reference are linear in 6 moments
//           sum(w), sum(w dx), sum(w dy), sum(w dx^2), sum(w dx dy), sum(w dy^2), w = G*dL/dpix,
//           accumulated in registers: no cross-lane reduction, and NO atomics -- each instance stores its
//           moment row to scratch at its UNSORTED list position (contiguous per Gaussian), which the
//           geometry backward then reduces in a fixed order.  Gradients are therefore bit-reproducible,
//           unlike the reference's float atomicAdd accumulation (RAS/backward.cu:562-572).
//           Workgroups are cut as 256 consecutive instances of the global sorted list: perfect balance.
//           Waves that straddle many sparse tiles switch to a per-lane gather of dL/dpix instead of
//           re-walking 256 pixels once per tile.
//           The reference's n_contrib skip (RAS/backward.cu:523-525) only prunes pairs that failed the
//           forward tests; re-evaluating the tests prunes the same pairs, so n_contrib is not needed.
#include "raster_state.hpp"

namespace r2 {

constexpr float ALPHA_MIN_2D = 0.00001f;   // RAS/forward.cu:374

// ------------------------------------------------------------------------------------------------ forward
template <bool NCONTRIB>
__global__ void __launch_bounds__(256) raster_render_forward_kernel(
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ chunk_base, const uint32_t *__restrict__ work_tile,
    uint32_t T, const uint32_t *__restrict__ point_list, const float4 *__restrict__ rec, int gx,
    float *__restrict__ partial, uint32_t *__restrict__ partial_last)
{
    const uint32_t w = blockIdx.x;
    if (w >= chunk_base[T]) return;
    const uint32_t tile = work_tile[w];
    const uint32_t j0 = (w - chunk_base[tile]) * FWD_CHUNK;
    const uint2 range = ranges[tile];
    const uint32_t beg = range.x + j0, end = min(range.y, beg + FWD_CHUNK);
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x;
    const float fx = (float)(tx * TILE2D + (tid & 15)), fy = (float)(ty * TILE2D + (tid >> 4));

    __shared__ float4 sA[256];
    __shared__ float2 sB[256];

    float C = 0.f;
    uint32_t last = 0;
    for (uint32_t base = beg; base < end; base += 256) {
        __syncthreads();
        const uint32_t k = base + tid;
        if (k < end) {
            const uint32_t id = point_list[k];
            const float4 a = rec[2 * id];
            const float2 b = *reinterpret_cast<const float2 *>(&rec[2 * id + 1]);
            sA[tid] = a;
            sB[tid] = b;
        }
        __syncthreads();
        const int n = min(256u, end - base);
#pragma unroll 4
        for (int j = 0; j < n; ++j) {
            const float4 a = sA[j];
            const float2 b = sB[j];
            const float dx = a.x - fx, dy = a.y - fy;
            const float p2 = dx * (a.z * dx + a.w * dy) + (b.x * dy) * dy;   // log2(e) * power
            const float alpha = b.y * __builtin_amdgcn_exp2f(p2);
            const bool ok = (p2 <= 0.0f) && (alpha >= ALPHA_MIN_2D);
            C += ok ? alpha : 0.f;
            if (NCONTRIB) last = ok ? (base - range.x) + (uint32_t)j + 1u : last;
        }
    }
    partial[(size_t)w * 256 + tid] = C;
    if (NCONTRIB) partial_last[(size_t)w * 256 + tid] = last;
}

// adds the partial sums of a tile's work items in list order and writes the image (zeros for empty tiles)
template <bool NCONTRIB>
__global__ void __launch_bounds__(256) raster_combine_kernel(
    const uint32_t *__restrict__ chunk_base, const float *__restrict__ partial,
    const uint32_t *__restrict__ partial_last, int W, int H, int gx, float *__restrict__ out_color,
    uint32_t *__restrict__ n_contrib)
{
    const uint32_t tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x;
    const int px = tx * TILE2D + (tid & 15), py = ty * TILE2D + (tid >> 4);
    const uint32_t w0 = chunk_base[tile], w1 = chunk_base[tile + 1];
    float C = 0.f;
    uint32_t last = 0;
    for (uint32_t w = w0; w < w1; ++w) {
        C += partial[(size_t)w * 256 + tid];
        if (NCONTRIB) {
            const uint32_t l = partial_last[(size_t)w * 256 + tid];
            last = l ? l : last;
        }
    }
    if (px < W && py < H) {
        out_color[py * W + px] = C;
        if (NCONTRIB) n_contrib[py * W + px] = last;
    }
}

// ------------------------------------------------------------------------------------------------ backward
__device__ __forceinline__ void pixel_moments(const float4 a, const float4 b, float dx, float bdy, float cdy2, float g,
                                              float &r0, float &r1, float &r3)
{
    const float p2 = dx * (a.z * dx + bdy) + cdy2;
    const float G = __builtin_amdgcn_exp2f(p2);
    const bool ok = (p2 <= 0.0f) && (b.y * G >= ALPHA_MIN_2D);
    const float w = ok ? G * g : 0.f;
    const float wdx = w * dx;
    r0 += w;
    r1 += wdx;
    r3 += wdx * dx;
}

// moments of w = G * dL/dpix over one tile whose 16x16 block of dL/dpix sits in this wave's LDS slab
// (zeros outside the image): every read is a wave-uniform b128 broadcast, the loop has no bounds logic.
__device__ __forceinline__ void tile_moments_uniform(const float4 a, const float4 b, const float4 *__restrict__ gt,
                                                     int x0, int y0, float *S)
{
    const float dx0 = a.x - (float)x0;
#pragma unroll 2
    for (int r = 0; r < TILE2D; ++r) {
        const float dy = a.y - (float)(y0 + r);
        const float bdy = a.w * dy;           // B2*dy
        const float cdy2 = (b.x * dy) * dy;   // C2*dy^2
        float g[TILE2D];
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const float4 v = gt[r * 4 + c4];
            g[4 * c4 + 0] = v.x; g[4 * c4 + 1] = v.y; g[4 * c4 + 2] = v.z; g[4 * c4 + 3] = v.w;
        }
        float r0 = 0.f, r1 = 0.f, r3 = 0.f;
#pragma unroll
        for (int c = 0; c < TILE2D; ++c) pixel_moments(a, b, dx0 - (float)c, bdy, cdy2, g[c], r0, r1, r3);
        S[0] += r0; S[1] += r1; S[3] += r3;
        S[2] += dy * r0; S[4] += dy * r1; S[5] += dy * dy * r0;
    }
}

// same, for a wave whose lanes sit in MANY different (sparse) tiles: every lane gathers the dL/dpix of its
// own tile straight from memory, one pass for the whole wave instead of one pass per tile.
__device__ __forceinline__ void tile_moments_gather(const float4 a, const float4 b, const float *__restrict__ dL, int W,
                                                    int H, int x0, int y0, float *S)
{
    const float dx0 = a.x - (float)x0;
    const int nrows = min(TILE2D, H - y0), ncols = min(TILE2D, W - x0);
    for (int r = 0; r < nrows; ++r) {
        const float dy = a.y - (float)(y0 + r);
        const float bdy = a.w * dy, cdy2 = (b.x * dy) * dy;
        const float *__restrict__ row = dL + (size_t)(y0 + r) * W + x0;
        float r0 = 0.f, r1 = 0.f, r3 = 0.f;
        for (int c = 0; c < ncols; ++c) pixel_moments(a, b, dx0 - (float)c, bdy, cdy2, row[c], r0, r1, r3);
        S[0] += r0; S[1] += r1; S[3] += r3;
        S[2] += dy * r0; S[4] += dy * r1; S[5] += dy * dy * r0;
    }
}

__global__ void __launch_bounds__(256) raster_render_backward_kernel(
    const uint32_t *__restrict__ tiles, const uint32_t *__restrict__ point_list, const uint32_t *__restrict__ perm,
    const float4 *__restrict__ rec, uint32_t R, int W, int H, int gx, uint32_t nchunks, const float *__restrict__ dL_dpix, float4 *__restrict__ part)
{
    __shared__ float4 gtile[4][64];   // one 16x16 dL/dpix block per wave
    const uint32_t chunk = xcd_remap(blockIdx.x, nchunks);
    if (chunk >= nchunks) return;
    const uint32_t k = chunk * 256u + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool live = k < R;
    uint32_t tile = 0xffffffffu, id = 0;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (live) {
        tile = tiles[k];
        id = point_list[k];
        a = rec[2 * id];
        b = rec[2 * id + 1];
    }
    float S[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
    float4 *gt = gtile[wave];

    // number of distinct tiles in this wave (the list is tile-sorted: count the run starts)
    const uint32_t prev_tile = __shfl_up(tile, 1);
    const unsigned long long heads = __ballot(live && (lane == 0 || tile != prev_tile));
    if (__popcll(heads) > 3) {
        if (live) tile_moments_gather(a, b, dL_dpix, W, H, (int)(tile % gx) * TILE2D, (int)(tile / gx) * TILE2D, S);
    } else {
        unsigned long long todo = __ballot(live);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t t = __builtin_amdgcn_readfirstlane(__shfl(tile, leader));
            const bool mine = live && tile == t;
            todo &= ~__ballot(mine);
            const int x0 = (int)(t % gx) * TILE2D, y0 = (int)(t / gx) * TILE2D;
            {   // stage the tile's dL/dpix: lane -> (row = lane/4, 4 columns), zero outside the image
                const int ry = y0 + (lane >> 2), cx = x0 + (lane & 3) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ry < H) {
                    const float *__restrict__ src = dL_dpix + (size_t)ry * W + cx;
                    if (cx + 3 < W && (W & 3) == 0) v = *reinterpret_cast<const float4 *>(src);
                    else {
                        if (cx + 0 < W) v.x = src[0];
                        if (cx + 1 < W) v.y = src[1];
                        if (cx + 2 < W) v.z = src[2];
                        if (cx + 3 < W) v.w = src[3];
                    }
                }
                __builtin_amdgcn_wave_barrier();   // readers of the previous tile are done (same wave, in order)
                gt[lane] = v;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (mine) tile_moments_uniform(a, b, gt, x0, y0, S);
        }
    }
    if (live) {
        // scratch row = the instance's position in the emission list (the sort carried it as payload)
        const uint32_t u = perm[k];
        part[2 * (size_t)u] = make_float4(S[0], S[1], S[2], S[3]);
        part[2 * (size_t)u + 1] = make_float4(S[4], S[5], 0.f, 0.f);
    }
}

// ------------------------------------------------------------------------------------------------ launchers
int launch_raster_render_forward(const RasterGeom &g, const RasterBinning &b, const RasterImage &im, int W, int H,
                                 float *out_color, bool write_ncontrib, hipStream_t s)
{
    const int gx = (W + TILE2D - 1) / TILE2D, gy = (H + TILE2D - 1) / TILE2D;
    const uint32_t T = (uint32_t)gx * gy;
    launch_build_work(im.ranges, T, FWD_CHUNK, im.chunk_base, im.work_tile, s);
    if (im.NW > 0) {
        if (write_ncontrib)
            raster_render_forward_kernel<true><<<dim3((unsigned)im.NW), dim3(256), 0, s>>>(
                im.ranges, im.chunk_base, im.work_tile, T, b.point_list, g.rec, gx, im.partial, im.partial_last);
        else
            raster_render_forward_kernel<false><<<dim3((unsigned)im.NW), dim3(256), 0, s>>>(
                im.ranges, im.chunk_base, im.work_tile, T, b.point_list, g.rec, gx, im.partial, im.partial_last);
    }
    if (write_ncontrib)
        raster_combine_kernel<true><<<dim3(T), dim3(256), 0, s>>>(im.chunk_base, im.partial, im.partial_last, W, H, gx,
                                                                  out_color, im.n_contrib);
    else
        raster_combine_kernel<false><<<dim3(T), dim3(256), 0, s>>>(im.chunk_base, im.partial, im.partial_last, W, H, gx,
                                                                   out_color, im.n_contrib);
    return 0;
}

int launch_raster_render_backward(const RasterGeom &g, const RasterBinning &b, int W, int H, size_t R,
                                  const float *dL_dpix, hipStream_t s)
{
    if (R == 0) return 0;
    const int gx = (W + TILE2D - 1) / TILE2D;
    const uint32_t nchunks = (uint32_t)((R + 255) / 256);
    const uint32_t grid = ((nchunks + 7u) >> 3) << 3;
    raster_render_backward_kernel<<<dim3(grid), dim3(256), 0, s>>>(b.tiles, b.point_list, b.perm, g.rec, (uint32_t)R, W, H, gx,
                                                                   nchunks, dL_dpix,
                                                                   reinterpret_cast<float4 *>(b.part));
    return 0;
}

}  // namespace r2
