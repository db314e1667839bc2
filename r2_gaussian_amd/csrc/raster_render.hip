// raster_render.hip -- per-tile additive line-integral render of the X-ray rasterizer and its backward.
//
// Reference: renderCUDA forward RAS/forward.cu:294-395, renderCUDA backward RAS/backward.cu:447-575.
// These two kernels are VALU/exp-bound (256 pixel-Gaussian pairs per 32 bytes gathered), so this file
// is compiled with FMA contraction ON; their results are tolerance-checked, not bit-checked.
//
// Forward : one workgroup = one 16x16 tile = 4 waves of 16x4 pixels; the tile's depth-sorted list is
//           staged through LDS in 256-record batches (32-byte packed records, two b128 gathers per
//           record), every lane then reads the records as wave-uniform LDS broadcasts.
// Backward: the loop nest is inverted.  One LANE owns one (tile, Gaussian) instance of the sorted list and
//           walks the 256 pixels of its tile; pixel data (dL/dpix) is wave-uniform and arrives through
//           scalar loads.  The 7 gradient terms of the reference are linear in 6 moments
//           sum(w), sum(w dx), sum(w dy), sum(w dx^2), sum(w dx dy), sum(w dy^2), w = G*dL/dpix,
//           accumulated in registers, so there is no cross-lane reduction and only 7 atomics per
//           INSTANCE instead of 7 per contributing (pixel, Gaussian) pair (RAS/backward.cu:562-572).
//           Workgroups are cut as 256 consecutive instances of the global sorted list, which balances
//           the load perfectly whatever the tile occupancy.
//           The reference's n_contrib skip (RAS/backward.cu:523-525) only prunes pairs that failed the
//           forward tests; re-evaluating the tests prunes the same pairs, so n_contrib is not needed.
#include "raster_state.hpp"

namespace r2 {

constexpr float ALPHA_MIN_2D = 0.00001f;   // RAS/forward.cu:374

template <bool NCONTRIB>
__global__ void __launch_bounds__(256) raster_render_forward_kernel(
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list, const float4 *__restrict__ rec, int W,
    int H, int gx, uint32_t T, float *__restrict__ out_color, uint32_t *__restrict__ n_contrib)
{
    const uint32_t tile = xcd_remap(blockIdx.x, T);
    if (tile >= T) return;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x;
    const int px = tx * TILE2D + (tid & 15), py = ty * TILE2D + (tid >> 4);
    const bool inside = px < W && py < H;
    const float fx = (float)px, fy = (float)py;
    const uint2 range = ranges[tile];

    __shared__ float4 sA[256];
    __shared__ float2 sB[256];

    float C = 0.f;
    uint32_t last = 0;
    for (uint32_t base = range.x; base < range.y; base += 256) {
        __syncthreads();
        const uint32_t k = base + tid;
        if (k < range.y) {
            const uint32_t id = point_list[k];
            const float4 a = rec[2 * id];
            const float4 b = rec[2 * id + 1];
            sA[tid] = a;
            sB[tid] = make_float2(b.x, b.y);
        }
        __syncthreads();
        const int n = min(256u, range.y - base);
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
    if (inside) {
        out_color[py * W + px] = C;
        if (NCONTRIB) n_contrib[py * W + px] = last;
    }
}

// moments of w = G * dL/dpix over one tile, for the instance held by this lane.  FULLW: the tile has all
// 16 columns, so the 16 wave-uniform dL loads of a row are unconditional (merged into wide scalar loads).
template <bool FULLW>
__device__ __forceinline__ void tile_moments(const float4 a, const float4 b, const float *__restrict__ dL_dpix, int W,
                                             int x0, int y0, int ncols, int nrows, float &S0, float &S1, float &S2,
                                             float &S3, float &S4, float &S5)
{
    const float dx0 = a.x - (float)x0;
    for (int r = 0; r < nrows; ++r) {
        const float dy = a.y - (float)(y0 + r);
        const float bdy = a.w * dy;           // B2*dy
        const float cdy2 = (b.x * dy) * dy;   // C2*dy^2
        const float *__restrict__ row = dL_dpix + (size_t)(y0 + r) * W + x0;
        float r0 = 0.f, r1 = 0.f, r3 = 0.f;
#pragma unroll
        for (int c = 0; c < TILE2D; ++c) {
            float g;   // wave-uniform address -> scalar load
            if (FULLW) g = row[c];
            else g = (c < ncols) ? row[min(c, ncols - 1)] : 0.f;
            const float dx = dx0 - (float)c;
            const float p2 = dx * (a.z * dx + bdy) + cdy2;
            const float G = __builtin_amdgcn_exp2f(p2);
            const bool ok = (p2 <= 0.0f) && (b.y * G >= ALPHA_MIN_2D);
            const float w = ok ? G * g : 0.f;
            const float wdx = w * dx;
            r0 += w;
            r1 += wdx;
            r3 += wdx * dx;
        }
        S0 += r0;
        S1 += r1;
        S3 += r3;
        S2 += dy * r0;
        S4 += dy * r1;
        S5 += dy * dy * r0;
    }
}

__global__ void __launch_bounds__(256) raster_render_backward_kernel(
    const uint64_t *__restrict__ keys, const uint32_t *__restrict__ point_list, const float4 *__restrict__ rec,
    uint32_t R, int W, int H, int gx, uint32_t nchunks, const float *__restrict__ dL_dpix,
    float *__restrict__ dL_dmean2D, float *__restrict__ dL_dconic, float *__restrict__ dL_dopacity,
    float *__restrict__ dL_dmu)
{
    const uint32_t chunk = xcd_remap(blockIdx.x, nchunks);
    if (chunk >= nchunks) return;
    const uint32_t k = chunk * 256u + threadIdx.x;
    const bool live = k < R;
    uint32_t tile = 0xffffffffu, id = 0;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (live) {
        tile = (uint32_t)(keys[k] >> 32);
        id = point_list[k];
        a = rec[2 * id];
        b = rec[2 * id + 1];
    }
    float S0 = 0.f, S1 = 0.f, S2 = 0.f, S3 = 0.f, S4 = 0.f, S5 = 0.f;

    // a wave usually sits inside one tile; at list boundaries it serves each tile in turn
    unsigned long long todo = __ballot(live);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t t = __builtin_amdgcn_readfirstlane(__shfl(tile, leader));
        const bool mine = live && tile == t;
        todo &= ~__ballot(mine);
        const int tx = t % gx, ty = t / gx;
        const int x0 = tx * TILE2D, y0 = ty * TILE2D;
        const int ncols = min(TILE2D, W - x0), nrows = min(TILE2D, H - y0);
        if (mine) {
            if (ncols == TILE2D)
                tile_moments<true>(a, b, dL_dpix, W, x0, y0, ncols, nrows, S0, S1, S2, S3, S4, S5);
            else
                tile_moments<false>(a, b, dL_dpix, W, x0, y0, ncols, nrows, S0, S1, S2, S3, S4, S5);
        }
    }
    if (live) {
        const float op = b.z, mu = b.w, opmu = b.y;
        const float A = a.z * (-2.0f * LN2), B = a.w * (-LN2), Cc = b.x * (-2.0f * LN2);   // undo the log2e pre-scale
        unsafeAtomicAdd(&dL_dmean2D[3 * id + 0], opmu * (0.5f * (float)W) * (-A * S1 - B * S2));
        unsafeAtomicAdd(&dL_dmean2D[3 * id + 1], opmu * (0.5f * (float)H) * (-Cc * S2 - B * S1));
        unsafeAtomicAdd(&dL_dconic[4 * id + 0], -0.5f * opmu * S3);
        unsafeAtomicAdd(&dL_dconic[4 * id + 1], -opmu * S4);
        unsafeAtomicAdd(&dL_dconic[4 * id + 3], -0.5f * opmu * S5);
        unsafeAtomicAdd(&dL_dopacity[id], mu * S0);
        unsafeAtomicAdd(&dL_dmu[id], op * S0);
    }
}

int launch_raster_render_forward(const RasterGeom &g, const RasterBinning &b, const RasterImage &im, int W, int H,
                                 float *out_color, bool write_ncontrib, hipStream_t s)
{
    const int gx = (W + TILE2D - 1) / TILE2D, gy = (H + TILE2D - 1) / TILE2D;
    const uint32_t T = (uint32_t)gx * gy;
    const uint32_t grid = ((T + 7u) >> 3) << 3;
    if (write_ncontrib)
        raster_render_forward_kernel<true><<<dim3(grid), dim3(256), 0, s>>>(im.ranges, b.point_list, g.rec, W, H, gx, T,
                                                                            out_color, im.n_contrib);
    else
        raster_render_forward_kernel<false><<<dim3(grid), dim3(256), 0, s>>>(im.ranges, b.point_list, g.rec, W, H, gx, T,
                                                                             out_color, im.n_contrib);
    return 0;
}

int launch_raster_render_backward(const RasterGeom &g, const RasterBinning &b, const RasterImage &im, int W, int H,
                                  size_t R, const float *dL_dpix, float *dL_dmean2D, float *dL_dconic,
                                  float *dL_dopacity, float *dL_dmu, hipStream_t s)
{
    (void)im;
    if (R == 0) return 0;
    const int gx = (W + TILE2D - 1) / TILE2D;
    const uint32_t nchunks = (uint32_t)((R + 255) / 256);
    const uint32_t grid = ((nchunks + 7u) >> 3) << 3;
    raster_render_backward_kernel<<<dim3(grid), dim3(256), 0, s>>>(b.keys, b.point_list, g.rec, (uint32_t)R, W, H, gx,
                                                                   nchunks, dL_dpix, dL_dmean2D, dL_dconic,
                                                                   dL_dopacity, dL_dmu);
    return 0;
}

}  // namespace r2
