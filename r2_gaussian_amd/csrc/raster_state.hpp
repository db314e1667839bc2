// raster_state.hpp -- private layout of the three opaque state buffers of the rasterizer
// (the reference's GeometryState / BinningState / ImageState, RAS/rasterizer_impl.h:33-67,
// RAS/rasterizer_impl.cu:155-192).  The layout is ours: 32-byte packed render records instead of
// three separate arrays, and tile ranges sized per TILE, not per pixel (survey quirk Q7).
#pragma once
#include "r2_common.hpp"

namespace r2 {

#ifndef R2_EXP_FWD_CHUNK
#define R2_EXP_FWD_CHUNK 1024
#endif
constexpr uint32_t FWD_CHUNK = R2_EXP_FWD_CHUNK;   // instances of one tile list rendered by one workgroup (load balance)
constexpr int PART_STRIDE = 8;        // floats per instance in the backward moment scratch (6 used)
constexpr float ALPHA_MIN_2D = 0.00001f;               // RAS/forward.cu:374
constexpr float LOG2_ALPHA_MIN_2D = -16.609640474436812f;   // log2(1e-5)
constexpr int SUB2D = 8;              // culling granularity of the render kernels: 8x8 pixel blocks of a 16x16 tile

// tile rectangle of a square of half-width `rad` around p (RAS/auxiliary.h:50-60); float->int truncation.  Used by the
// preprocess / duplicate kernels and by the render backward, which recomputes an instance's emission index from it.
__device__ __forceinline__ void tile_rect(float px, float py, int rad, int gx, int gy, int &x0, int &y0, int &x1, int &y1)
{
    x0 = min(gx, max(0, (int)((px - rad) / TILE2D)));
    y0 = min(gy, max(0, (int)((py - rad) / TILE2D)));
    x1 = min(gx, max(0, (int)((px + rad + TILE2D - 1) / TILE2D)));
    y1 = min(gy, max(0, (int)((py + rad + TILE2D - 1) / TILE2D)));
}

// Batched views.  V views of the same Gaussians are rendered as ONE scene of V * P "view instances" (id = v * P + i) on a
// tile grid that stacks the views' grids: tile id = (v * gy + ty) * gx + tx.  Everything between the preprocess and the
// render kernels (depth order, instance emission, tile sort, ranges, work lists) is oblivious to it; records keep their
// per-view pixel coordinates, so a batched view is evaluated with exactly the arithmetic of a single one.  V = 1 is the
// reference's call.
// MV = false: single view, compiled without the extra division (the kernels are instantiated for both).
template <bool MV>
__device__ __forceinline__ void tile_decode(uint32_t tile, int gx, int gy, int &tx, int &ty, int &v)
{
    tx = (int)(tile % (uint32_t)gx);
    const int tyt = (int)(tile / (uint32_t)gx);
    v = MV ? tyt / gy : 0;
    ty = tyt - v * gy;
}

__device__ __forceinline__ int row_tier(float A2, float L, float hx)
{
    // 0: recurrence over the whole 8-pixel row; 1: recurrence re-anchored every 4 pixels (3 steps: safe down to a
    // conditional sigma of ~0.33 px); 2: exact per-pixel evaluation
    // q(c) = log2 G(c) = p(c) - L <= 0 is the concave parabola; the evaluated exponent p = q + L underflows (exp2 -> 0, or a
    // flushed denormal) when q0 < -(126 + L), and a later pixel still passes the cut-off when q(c) >= log2(1e-5) - L.
    // sqrt(-q) is Lipschitz along a pixel row with constant sqrt|A2|, so c steps from an underflowed start cannot reach the
    // cut-off if sqrt(126 + L) - c sqrt|A2| >= sqrt(L - log2(1e-5) + 1).  (Round 1 used sqrt(126) for the first term: too
    // optimistic by sqrt(126) - sqrt(126 + L) for L < 0 -- found by the pure-1e-4 parity check on sub-pixel Gaussians, which
    // lost one 1.3e-5 contribution.)  The 0.5 keeps clear of the last normal binade.
    // These thresholds are also where the recurrence's ACCURACY ends: carrying alpha * 2^64 through the recurrence moves the
    // underflow out of reach (measured, round 2) and would allow |A2| <= ~2 on the 8-step path, but 7 ratio steps at
    // |A2| > ~1.1 put the covariance gradients of such Gaussians 1.3-2.6x outside the 1e-4 bound.
    // v_sqrt_f32 (1 ulp) is plenty for a threshold with this much margin; the correctly rounded sqrtf() expands to ~20
    // instructions each, and this runs once per entry per 64-entry step in both render kernels (measured: +1.4 us each)
    const float room = __builtin_amdgcn_sqrtf(fmaxf(125.5f + fminf(L, 0.f), 0.f)) - __builtin_amdgcn_sqrtf(fmaxf(L - LOG2_ALPHA_MIN_2D, 0.f) + 1.0f);
    const float s8 = room * (1.0f / 7.0f), s4 = room * (1.0f / 3.0f);
    const float a = fabsf(A2);
    if (!(hx < 3.0e38f) || !(room > 0.f)) return 2;
    return a <= s8 * s8 ? 0 : (a <= s4 * s4 ? 1 : 2);
}
__device__ __forceinline__ bool needs_exact_row(float A2, float L, float hx) { return row_tier(A2, L, hx) == 2; }
// Round 6, the tier with the recurrence across the rows (fwd_item): the walks start at (row 4 | 3, column 0) of the block, go up to
// three steps along the column and then seven along a row.  With q = -d^T M d (M positive definite), sqrt(-q) is the norm
// |M^(1/2) d|: between the anchor and a pixel at displacement v it changes by at most sqrt(v^T M v) <= sqrt(49 |A2| + 9 |C2| +
// 21 |B2|), so an underflowed anchor cannot hide a pixel above the cut-off if that stays below row_tier's `room` (for an isotropic
// Gaussian: sigma >= 0.92 px, against 0.85 px for the seven row steps alone; the bound 7 sqrt|A2| + 3 sqrt|C2| that ignores how
// the two displacements combine sent sigma < 1.2 px -- a third of the benchmark cloud's entries -- down the exact path).  A row
// start the column walk underflowed is covered by the row criterion, which this implies.  0: both recurrences; 2: exact
// per-pixel evaluation.  (Without the row recurrence -- R2_EXP_NO_YRECUR -- this is row_tier.)
__device__ __forceinline__ int item_tier(float A2, float B2, float C2, float L, float hx, float hy)
{
#ifdef R2_EXP_NO_YRECUR
    (void)B2; (void)C2; (void)hy;
    return row_tier(A2, L, hx);
#else
    const float room = __builtin_amdgcn_sqrtf(fmaxf(125.5f + fminf(L, 0.f), 0.f)) - __builtin_amdgcn_sqrtf(fmaxf(L - LOG2_ALPHA_MIN_2D, 0.f) + 1.0f);
    if (!(hx < 3.0e38f) || !(hy < 3.0e38f) || !(room > 0.f)) return 2;
    const float reach2 = 49.0f * fabsf(A2) + 9.0f * fabsf(C2) + 21.0f * fabsf(B2);
    return reach2 <= room * room ? 0 : 2;
#endif
}

// does the bounding box (px +- hx, py +- hy) of a Gaussian's alpha >= 1e-5 region touch the pixel block
// [x0, x0+n) x [y0, y0+n)?  (pixel centres are the integers; +-inf half-extents mean never / always)
__device__ __forceinline__ bool block_live(float px, float py, float hx, float hy, float x0, float y0, float n)
{
    return (px - hx <= x0 + (n - 1.0f)) && (px + hx >= x0) && (py - hy <= y0 + (n - 1.0f)) && (py + hy >= y0);
}

// the four 8x8 blocks of tile (tx, ty) that the bounding box touches, as a mask: bit q = block (q % 2, q / 2).  Evaluated ONCE per
// (tile, Gaussian) instance where the instance is emitted (round 6: the tile-first scatter kernel holds the Gaussian's record in
// registers anyway) and carried through the per-tile sort in the low bits of the id word, so that the render kernels -- forward and
// backward -- read a block's live entries off the sorted list instead of gathering every record to test it again.
__device__ __forceinline__ uint32_t block_mask4(float px, float py, float hx, float hy, int tx, int ty)
{
    const float x0 = (float)(tx * TILE2D), y0 = (float)(ty * TILE2D);
    uint32_t m = 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (block_live(px, py, hx, hy, x0 + (float)((q & 1) * SUB2D), y0 + (float)((q >> 1) * SUB2D), (float)SUB2D)) m |= 1u << q;
    return m;
}
constexpr int MASK_BITS = 4;   // masked list entry = id << MASK_BITS | block mask (ids below 2^28)
// number of set bits of a ballot below this lane (v_mbcnt: no 64-bit lane mask to keep in registers)
__device__ __forceinline__ uint32_t ballot_rank(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// (An exact second stage -- does the ELLIPSE {alpha >= 1e-5}, not its bounding box, reach the block? 40.5 % instead of 46 %
// of the (entry, block) pairs -- was built and measured in round 2: parity green, forward 47.5 -> 50.9 us (the test costs more
// than the dropped entries save), backward unchanged (its rounds are quantised: 118 or 104 items per chunk are both 2 rounds).)

// tile-first: how the Gaussians are dealt to the producer workgroups (the preprocess and the scatter kernel share the mapping:
// workgroup w owns Gaussians [w * per_wg, (w + 1) * per_wg), thread t of it the Gaussians t, t + threads, ... of that range).
// Round 4 used 1024 Gaussians = 1024 threads per workgroup whatever P: 300k Gaussians were 294 workgroups on 256 CUs, 38 CUs ran
// two of them and the kernel ended when THEY did (in-kernel stamps: median workgroup 12.3 us, slowest 19.1 us); 50k Gaussians were
// 49 workgroups on 256 CUs.  Now the grid is a whole number of workgroups per CU -- one per CU (up to 2048 Gaussians each, two
// per thread) where that is enough -- and small clouds get more, smaller workgroups.  Few workgroups also means few same-address
// atomics per tile counter (they retire at ~90 per microsecond device-wide).
constexpr uint32_t TF_THREADS_MAX = 1024, TF_PER_THREAD_MAX = 2, TF_PER_WG_MIN = 256;
// Depth slabs (round 5).  A tile list longer than one sort workgroup holds (8192 entries) is sorted by several workgroups that each
// read the WHOLE list to pick out their depth range -- on trained, densified clouds (lists of 23 k entries) that is most of the sort
// kernel's time.  When the previous call on this detector saw such lists, the chain bins by LIST = tile * slabs + slab instead of by
// tile: slab = a monotone function of the depth key laid over the previous call's key range (keys outside it clamp into the end
// slabs), slabs in {1, 2, 4}.  Monotone => a tile's slabs, one after the other, each sorted on (key, id), ARE the tile's list in
// (depth, id) order, whatever the range and the count: the choice affects speed only.
constexpr uint32_t TF_MAX_SLABS = 4;
constexpr uint32_t TF_SLAB_SPLIT_ABOVE = 12288;  // longest tile list beyond which the lists are cut into slabs (8192 = what one sort
                                                 // workgroup holds; up to ~two parts the several-workgroups-per-list path is cheaper)
struct TFSlabs { uint32_t n, lo; float scale; };
__host__ __device__ __forceinline__ uint32_t tf_slab_of(uint32_t key, const TFSlabs &sl)
{
    const uint32_t k = key > sl.lo ? key - sl.lo : 0u;
    const uint32_t v = (uint32_t)((float)k * sl.scale);
    return v < sl.n - 1u ? v : sl.n - 1u;
}
struct TFGrid { uint32_t wgs, per_wg, threads; };
inline TFGrid tf_grid(int P, int cus)
{
    const unsigned long long cap = (unsigned long long)TF_THREADS_MAX * TF_PER_THREAD_MAX, c = (unsigned long long)(cus > 0 ? cus : 1);
    const unsigned long long k = ((unsigned long long)P + c * cap - 1) / (c * cap);            // workgroups per CU
    unsigned long long per = ((unsigned long long)P + k * c - 1) / (k * c);
    if (per < TF_PER_WG_MIN) per = TF_PER_WG_MIN;
    // beyond one workgroup per CU (> 524k Gaussians) the kernels run in rounds anyway, and rounds of plain one-Gaussian-per-thread
    // workgroups overlap their phases better than half as many doing everything twice (1M Gaussians / 1024^2: scatter 50.6 us with
    // 977 workgroups of 1024, 62 us with 512 of 1954; the preprocess 51 vs 47 -- measured, round 5)
    if (k > 1) per = TF_THREADS_MAX;
    TFGrid g;
    g.per_wg = (uint32_t)per;
    g.wgs = (uint32_t)(((unsigned long long)P + per - 1) / per);
    const unsigned long long th = (per + 63) / 64 * 64;
    g.threads = (uint32_t)(th > TF_THREADS_MAX ? TF_THREADS_MAX : th);
    return g;
}
struct RasterGeom {
    float4 *rec;              // [2P]  {px, py, A2, B2} {C2, L, hx, hy}: A2,B2,C2 = conic * (-log2e/2, -log2e, -log2e/2),
                              //       L = log2(opacity*mu), (hx, hy) = half-extents of the alpha >= 1e-5 bounding box
    float2 *op_mu;            // [P]   {opacity, mu} for the geometry backward
    uint32_t *depth_key;      // [P]   bits of view-space z (positive floats order like unsigned ints); 0xFFFFFFFF = culled
    uint32_t *iota;           // [P]   0..P-1, value input of the depth sort
    uint32_t *depth_sorted;   // [P]   sorted depth keys (unused afterwards)
    uint32_t *order;          // [P]   Gaussian ids in (depth, id) order; culled ones last
    uint32_t *first;          // [P]   index of the Gaussian's first instance in the unsorted (emission) list
    float *cov3D;             // [6P]
    uint32_t *tiles_touched;  // [P]
    uint32_t *host_words;     // [DW_COUNT] the words the host reads back (num_rendered, overflow flag, "thin Gaussians present"
                              //       flag, key extrema, visible count): the control block at the start of psort_temp
    uint32_t *offsets;        // [P]   inclusive scan of tiles_touched[order[j]]: instance runs in depth order
    char *scan_temp;
    size_t scan_bytes;
    char *dorder_temp;        // depth order (bucket sort) workspace; starts with the control block = host_words
    size_t dorder_bytes;
    char *psort_temp;
    size_t psort_bytes;
    // tile-first binning (raster_tilefirst.hip) only, carved BEHIND everything else so that the layout the backward and the
    // introspection compute from P alone is unchanged: the Gaussian's tile rectangle (depth_rect_pack) and, per producer
    // workgroup (tf_grid) and tile, the offset of that workgroup's instances inside the tile's list
    uint32_t *tf_rect;        // [P]
    uint32_t *tf_wgoff;       // [producer workgroups][lists = T x slabs]
    uint32_t *tf_wgmm;        // [producer workgroups][2] key range {max, ~min} of every producer workgroup
    size_t bytes;
    static RasterGeom carve(char *chunk, int P, size_t tf_T = 0, size_t tf_wgs = 0)
    {
        RasterGeom g;
        Bump b(chunk);
        g.rec = b.take<float4>(2 * (size_t)P);
        g.op_mu = b.take<float2>(P);
        g.depth_key = b.take<uint32_t>(P);
        g.iota = b.take<uint32_t>(P);
        g.depth_sorted = b.take<uint32_t>(P);
        g.order = b.take<uint32_t>(P);
        g.first = b.take<uint32_t>(P);
        g.cov3D = b.take<float>(6 * (size_t)P);
        g.tiles_touched = b.take<uint32_t>(P);
        g.offsets = b.take<uint32_t>(P);
            g.scan_bytes = scan_gather_temp_bytes(P);
        g.scan_temp = b.take<char>(g.scan_bytes);
        g.dorder_bytes = depth_order_temp_bytes((size_t)P);
        g.dorder_temp = b.take<char>(g.dorder_bytes);
        g.host_words = chunk ? depth_order_words(g.dorder_temp, (size_t)P) : nullptr;
        g.psort_bytes = sort_temp_bytes((size_t)P);   // radix fallback of the depth order (kept apart: the control block at
        g.psort_temp = b.take<char>(g.psort_bytes);   // the start of dorder_temp must survive until the backward)
        g.tf_rect = b.take<uint32_t>(tf_T ? (size_t)P : 0);
        g.tf_wgoff = b.take<uint32_t>(tf_wgs * tf_T);
        g.tf_wgmm = b.take<uint32_t>(tf_wgs * 2);
        g.bytes = b.total();
        return g;
    }
};

struct RasterBinning {
    uint32_t *tiles_unsorted; // [R]  tile id of every instance, emitted Gaussian by Gaussian in depth order
    uint32_t *tiles;          // [R]  tile id of every SORTED instance (written by the multi-pass sort, or filled from the
                              //      ranges at the start of the backward when the single-pass sort skipped the key scatter)
    uint32_t *masked;         // [R]  round 6: point_list[k] << MASK_BITS | block_mask4 of instance k (what the render kernels read)
    uint32_t *vals_unsorted;  // [R]  Gaussian id of every instance (emission order)
    uint32_t *inv;            // [R]  sorted position of emission index u (inverse permutation of the tile sort; kept for
                              //      introspection -- the backward recomputes emission indices arithmetically)
    uint32_t *point_list;     // [R]  sorted Gaussian ids (the sort's second payload; the reference's point_list, bit-identical)
    float *part;              // [R*PART_STRIDE] backward scratch: per-instance moments, indexed by EMISSION position
                              //      (a Gaussian's rows are contiguous: [first, first + tiles_touched))
    char *sort_temp;
    size_t sort_bytes;
    size_t bytes;
    static RasterBinning carve(char *chunk, size_t R)
    {
        RasterBinning s;
        Bump b(chunk);
        // Order matters (round 4): the tile-first forward sizes this buffer by a PREDICTED instance count and launches its
        // kernels before the host knows R, while the backward carves it with the true R.  What crosses from forward to backward
        // must therefore sit where both agree: point_list at offset 0, and tiles right behind it, at align128(4 R) -- an
        // address the forward's last kernel computes ON THE DEVICE from the R it reads there (binning_tiles_ptr).  Everything
        // else is scratch of one side only.
        s.point_list = b.take<uint32_t>(R);
        s.tiles = b.take<uint32_t>(R);
        s.masked = b.take<uint32_t>(R);
        s.tiles_unsorted = b.take<uint32_t>(R);
        s.vals_unsorted = b.take<uint32_t>(R);
        s.inv = b.take<uint32_t>(R);
        s.part = b.take<float>(R * PART_STRIDE);
        s.sort_bytes = sort_temp_bytes(R);
        s.sort_temp = b.take<char>(s.sort_bytes);
        s.bytes = b.total();
        return s;
    }
};

// bin.tiles of a binning buffer carved with R instances, from its base (see RasterBinning::carve; Bump aligns to 128 bytes)
__host__ __device__ __forceinline__ uint32_t *binning_tiles_ptr(char *base, size_t R)
{
    return reinterpret_cast<uint32_t *>(base + (((R * sizeof(uint32_t)) + 127) & ~size_t(127)));
}
__host__ __device__ __forceinline__ uint32_t *binning_masked_ptr(char *base, size_t R)   // ... and bin.masked, the third array
{
    return reinterpret_cast<uint32_t *>(base + 2 * (((R * sizeof(uint32_t)) + 127) & ~size_t(127)));
}

constexpr uint32_t TF_SMALL_CAP = 1536;     // tile-first: tile lists beyond this many entries get a whole sort workgroup (raster_tilefirst.hip)
constexpr uint32_t TF_MAX_TILES = 4096;     // tile-first: per-workgroup LDS histogram over the tiles
constexpr int TF_NOT_TAKEN = -1000001;      // raster_forward_tilefirst: nothing launched, run the general chain
constexpr uint32_t TF_MARK = 0x71FEu;       // host word DW_PMAX of a forward that took the tile-first path (introspection)
// Counters of the tile-first path that several workgroups of one kernel bump with atomics.  They live in a small persistent
// allocation per (host thread, device, stream) and are SELF-RESETTING: whoever consumes a counter last puts the zero back, so no
// zero-fill launch precedes the forward (a launch boundary costs 3-4 us).  All zero between calls.
struct TFCounters {
    unsigned long long total;   // (visible Gaussians << 40) | instances handed out so far (a workgroup's base = the low bits)
    uint32_t unused;
    uint32_t thin;              // a Gaussian needs the re-anchored row recurrence (row_tier == 1)
    uint32_t pad[12];
    uint32_t tile_count[TF_MAX_TILES];   // instances per tile
};

struct RasterImage {
    uint2 *ranges;         // [T]
    uint32_t *chunk_base;  // [T+2] exclusive scan of ceil(len/FWD_CHUNK): first work item of each tile; [T] = total,
                           //       [T+1] = total + empty tiles (appended to the work list when the combine is fused)
    uint32_t *tile_done;   // [4T] per (tile, 8x8 block): work items that have stored their partial sums (fused combine)
    uint4 *work_tile;      // [NW]  tile of each forward work item
    float *partial;        // [NW*256] per-work-item partial pixel sums, combined in list order
    uint32_t *partial_last;// [NW*256] debug only: last contributing list position inside the chunk
    uint32_t *n_contrib;   // [N]  last contributing list position per pixel; only written in debug mode
    char *work_temp;       // scratch of the parallel work-list construction (only for > 4096 tiles)
    uint4 *tf_parts;       // [NP + lists] tile-first only (forward scratch): the sort kernel's work lists -- NP "big" parts {tile,
                           //      part | parts << 16, first instance, instances}, then up to T short lists {tile, 0, first, instances}
    size_t NP;             // upper bound on big parts: every one stands for > TF_SMALL_CAP instances
    size_t NW;             // upper bound on work items: R/FWD_CHUNK + T
    size_t bytes;
    // everything the backward or the introspection reads (ranges, chunk_base) sits at offsets that depend on T only
    static RasterImage carve(char *chunk, size_t T, size_t N, size_t R, bool debug, size_t tf_lists = 0)
    {
        RasterImage s;
        Bump b(chunk);
        s.NW = R / FWD_CHUNK + T;
        s.ranges = b.take<uint2>(T);
        s.chunk_base = b.take<uint32_t>(T + 2);
        s.tile_done = b.take<uint32_t>(4 * T);
        s.work_tile = b.take<uint4>(s.NW);
        s.partial = b.take<float>(s.NW * 256);
        s.partial_last = b.take<uint32_t>(debug ? s.NW * 256 : 0);
        s.n_contrib = b.take<uint32_t>(debug ? N : 0);
        s.work_temp = b.take<char>(build_work_temp_bytes(T));
        s.NP = tf_lists ? R / TF_SMALL_CAP + 1 : 0;
        s.tf_parts = b.take<uint4>(tf_lists ? s.NP + tf_lists : 0);
        s.bytes = b.total();
        return s;
    }
};

// launchers (raster_geom.hip, raster_render.hip)
int launch_raster_preprocess(const RasterGeom &g, int P /* per view */, int V, const float *means3D, const float *scales, float scale_modifier,
                             const float *rotations, const float *opacities, const float *cov3D_precomp,
                             const float *view, const float *proj, int W, int H, float tan_fovx, float tan_fovy,
                             int mode, int *radii, uint32_t *thin_flag, const DepthReg &reg, bool store_cov3D, hipStream_t s);
// tile-first binning, first kernel: the preprocess + per-tile instance counts + every Gaussian's run of scratch rows.  The totals
// stay in the counters: the scatter kernel's workgroup 0 posts them to the state's host words and to the mailbox
int launch_raster_preprocess_tf(const RasterGeom &g, int P /* per view */, int V, const TFGrid &grid, const TFSlabs &slabs, const float *means3D, const float *scales,
                                float scale_modifier, const float *rotations, const float *opacities, const float *cov3D_precomp,
                                const float *view, const float *proj, int W, int H, float tan_fovx, float tan_fovy, int mode,
                                int *radii, TFCounters *ctr, hipStream_t s);
// 40 bits of instances (P < 2^24 Gaussians of <= 2^16 tiles each), 24 of visible Gaussians; a total beyond the 31-bit num_rendered
// of the API reaches the host as an out-of-range word, which it rejects
__host__ __device__ __forceinline__ uint32_t tf_total_instances(unsigned long long tot)
{
    const unsigned long long inst = tot & ((1ull << 40) - 1ull);
    return inst > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)inst;
}
int launch_raster_duplicate(const RasterGeom &g, const RasterBinning &b, int P /* per view */, int V, const int *radii, int W, int H,
                            const uint32_t *nvis /* device word: visible prefix of order/offsets, or null = all P */,
                            hipStream_t s);
// emission from the slab depth order's sorted records (hinted path; nvis = device word with the number of visible instances)
int launch_raster_duplicate_sorted(const RasterGeom &g, const RasterBinning &b, int P, int V, int W, int H, const uint32_t *nvis,
                                   hipStream_t s);
// the same by output range, one workgroup per sort tile, which also leaves the tile sort's histograms (plan from tile_sort_plan);
// false = not applicable here (nothing launched: use launch_raster_duplicate_sorted and the sort's own histogram pass)
bool launch_raster_emit_hist(const RasterGeom &g, const RasterBinning &b, int P, int V, int W, int H, const uint32_t *nvis, size_t R,
                             const TileSortPlan &plan, hipStream_t s);
int launch_mark_visible(int P, const float *means3D, const float *view, uint8_t *present, hipStream_t s);
int launch_raster_geom_backward(int P /* per view */, int V, const float *means3D, const int *radii, const float *cov3D, const float *scales,
                                const float *rotations, float scale_modifier, int W, int H, float tan_fovx,
                                float tan_fovy, const float *view, const float *proj, float *dL_dconic,
                                float *dL_dmu, float *dL_dmean2D, float *dL_dopacity, float *dL_dmean3D,
                                float *dL_dcov3D, float *dL_dscale, float *dL_drot, int mode, const RasterGeom &g,
                                const float *part, hipStream_t s);
// tf_bin_base / tf_words (tile-first forward only): fill_tiles is then computed ON THE DEVICE as binning_tiles_ptr(tf_bin_base,
// tf_words[DW_TOTAL]) -- the binning buffer was carved with a predicted count, the backward will carve it with the true one
int launch_raster_render_forward(const RasterGeom &g, const RasterBinning &b, const RasterImage &im, int W, int H, int V,
                                 float *out_color, bool write_ncontrib, uint32_t *fill_tiles, bool fused_combine,
                                 hipStream_t s, char *tf_bin_base = nullptr, const uint32_t *tf_words = nullptr,
                                 size_t view_instances = 0 /* P x V: ids of the masked list */);
// the one-wave forward kernel is in use (R2_FWD_WAVE=0: the four-wave kernel of rounds 1-5); it takes its work list longest first,
// the four-wave kernel in tile order (WorkListOut::longest_first)
bool raster_forward_wave_kernel_on();
inline bool raster_ids_leave_room_for_masks(size_t view_instances) { return view_instances < ((size_t)1 << (32 - MASK_BITS)); }
// raster_tilefirst.hip
int raster_forward_tilefirst(const char *what, r2_alloc_fn geometryBuffer, void *geometry_user, r2_alloc_fn binningBuffer,
                             void *binning_user, r2_alloc_fn imageBuffer, void *image_user, int P, int V, int width, int height,
                             const float *means3D, const float *opacities, const float *scales, float scale_modifier,
                             const float *rotations, const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix,
                             float tan_fovx, float tan_fovy, int mode, float *out_color, int *radii, hipStream_t s);
// a finished forward's instance count and depth-key range: the next call's prediction
void raster_tilefirst_note(int P, int V, int W, int H, uint32_t num_rendered, bool thin, uint32_t kmax, uint32_t kmin);
void raster_tilefirst_release();   // the calling thread's counters and predictions (r2_thread_release)
int raster_resolve_deferred(const char *what, int token, hipStream_t s, uint32_t *num_rendered);   // see r2_defer_count_control
int launch_raster_render_backward(const RasterGeom &g, const RasterBinning &b, const int *radii, int W, int H, int V, size_t R,
                                  const float *dL_dpix, hipStream_t s, size_t view_instances = ~(size_t)0 /* P x V */);

}  // namespace r2
