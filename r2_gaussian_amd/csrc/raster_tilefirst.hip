// raster_tilefirst.hip -- tile-first binning of the X-ray rasterizer (rounds 4-5): the reference's
// duplicateWithKeys -> SortPairs(tile | depth) -> identifyTileRanges (RAS/rasterizer_impl.cu:70-138,275-316) as
//     1. raster_preprocess_tf_kernel (raster_geom.hip)   preprocess + per-list instance counts (LDS histogram per workgroup, one
//                                                        returning global atomic per (workgroup, list)); the totals stay in the
//                                                        counters.  One workgroup per CU, up to two Gaussians per thread (tf_grid)
//     2. raster_tf_scatter_kernel                        every instance straight into its LIST's segment, as a (depth key,
//                                                        Gaussian id) pair, in arbitrary order inside the segment; three service
//                                                        workgroups meanwhile post the totals to the host and build tile ranges,
//                                                        the render work list and the sort parts
//     3. raster_tf_sort_kernel                           one 256-thread group per short list, one workgroup per long one (very
//                                                        long ones: several, by depth range) sorts its segment by (depth key, id)
//                                                        in LDS and writes point_list
// list = tile, or tile x depth slab when the lists are long (TFSlabs, raster_state.hpp).
// A stable sort by tile followed by a sort of every tile's entries on (depth bits, id) IS the reference's (tile | depth) order
// with its tie rule: point_list and ranges are bit-identical to the global-depth-order chain (raster_api.hip), which stays as
// the general path (first call on a detector size, batched views, debug mode, grids of more than 4096 tiles).
//
// Why (profiles/r03e_step_timeline.md): the global depth order costs the forward six launches -- zero-fill,
// preprocess with one 64-bit bucket atomic per Gaussian, dual scan x 2, place, rank -- before a single instance is emitted, and
// every launch boundary on this part is 3-5 us.  Here the forward is preprocess -> scatter -> sort -> render: no zero-fill (the
// counters are self-resetting), no scan kernel (every scatter workgroup scans the <= 4096 list counts itself), no radix pass.
// Round 5 (DESIGN.md section 4 has the stamps): these kernels are ONE round of workgroups each, so their time is the life of the
// slowest workgroup plus whatever follows it -- hence one producer workgroup per CU, no epilogue in the preprocess (the scatter's
// workgroup 0 posts the totals), the scatter's serial jobs on workgroups of their own.  The two-kernel form of this chain (no
// scatter: producers write their own instances grouped by list, the sort gathers a list from ~300 runs) was built and measured
// slower: profiles/experiments/r05_tilefirst_two_kernel_chain.patch.
//
// The host round trip.  num_rendered sizes the binning / image state (the reference's D2H, RAS/rasterizer_impl.cu:279).  The old
// chain hides it behind place + rank; here nothing is left to hide it behind, so the two buffers are sized by a PREDICTION (the
// largest count of the thread's recent calls with this P, + 25 %; a P it has not rendered yet -- every densification -- is seeded
// from its last call on the same detector) and kernels 2-4 are enqueued at once; they read the true count on the device and do
// nothing when it exceeds the prediction, in which case the host -- which reads the count from the mailbox as before -- sizes the
// buffers exactly and enqueues them again.  Results never depend on the prediction.
// What must survive from forward to backward sits at prediction-independent offsets (RasterBinning::carve).
#include "raster_state.hpp"
#include "dispatch.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <vector>

R2_TS_DEFINE(tilefirst)

namespace r2 {

namespace {

constexpr int TFS_THREADS = (int)TF_THREADS_MAX;   // scatter: always full workgroups (the scan of the tile counts uses all of them);
                                                   // the producer's mapping (tf_grid) says which threads own Gaussians
// ---- the sort kernel: 1024-thread workgroups of two kinds
//   "big"    one tile list of > TFK_SMALL_CAP entries (or one depth range of a list beyond TFK_BIG_CAP) sorted by the whole workgroup;
//   "group"  TFK_GROUPS lists of <= TFK_SMALL_CAP entries, one per 256-thread part of the workgroup, in lockstep (same code,
//            common barriers, own slice of the LDS).
// What was measured on the way (300k / 512^2, 1.16 M instances, list lengths 0 .. 8776, median 49; in-kernel stamps, -DR2_EXP_TS):
//  * one workgroup per <= 4096-entry PART of a list, every part histogramming the whole list to find its depth range: 25 us --
//    the dense tiles hold most of the instances and were read 2-3 x S times;
//  * whole lists in one workgroup: 18 us with 1024 threads / <= 8192 entries (100 KB of LDS, one workgroup per CU), 22 us with
//    512 threads / <= 4096 (three per CU, more lists in parts).  A workgroup's life is 7 us of dependent round trips -- counts
//    -> descriptor -> entries, all written by the previous kernel on other XCDs, i.e. served from memory -- then 3.5 us of
//    sorting; keeping the entries in registers, one entry per bucket (no rank reads), branch-free prefetch of every load and
//    agent-scope stores in the scatter kernel (data past the L2s before the boundary: scatter + 9 us, loads unchanged) moved
//    nothing: the kernel is those round trips times the number of rounds.
constexpr int TFK_THREADS = 1024;
constexpr int TFK_GROUPS = TFK_THREADS / 256;
// (the waves of a workgroup share ONE LDS pipe: entries stay in registers from the global load to the placement, buckets are as
//  many as entries -- most hold one, whose rank needs no read at all)
constexpr uint32_t TFK_SMALL_CAP = 1536, TFK_SMALL_PER = 6, TFK_SMALL_BINS = 1024;     // per 256-thread group
constexpr uint32_t TFK_BIG_CAP = 8192, TFK_BIG_PER = 8, TFK_BIG_BINS = 2048;           // whole workgroup (78 KB of LDS, <= 64 VGPRs: two per CU)
constexpr uint32_t TFK_BIG_TARGET = 5120;   // lists beyond TFK_BIG_CAP: ceil(n / 5120) parts, each a range of the list's depth histogram
constexpr uint32_t TFK_COARSE = 1024;       // bins of that histogram (one per thread)
constexpr size_t TFK_LDS = TFK_BIG_CAP * sizeof(unsigned long long) + (TFK_BIG_BINS + 1 + TFK_COARSE + 1) * sizeof(uint32_t);
static_assert(TFK_GROUPS * TFK_SMALL_CAP * sizeof(unsigned long long) + TFK_GROUPS * (TFK_SMALL_BINS + 1) * sizeof(uint32_t) <= TFK_LDS,
              "the groups fit the big layout");
static_assert(TFK_SMALL_CAP == TF_SMALL_CAP && TFK_COARSE == TFK_THREADS && TFK_BIG_CAP >= TFK_BIG_TARGET + TFK_BIG_TARGET / 2,
              "parts need slack over their target size");

// ---- 2. scatter.  The first TFS_SERVICE workgroups do not scatter; while the others run they (0) post the call's totals to the
// host, (1) build the tile ranges and the render kernel's work list, (2) build the sort kernel's work lists from the tile counts --
// one workgroup each: as ONE workgroup's serial job (round 4, 9 us) they became the kernel's tail once the producers got faster
constexpr uint32_t TFS_SERVICE = 3;
template <bool SL /* depth slabs in use */>
__global__ void __launch_bounds__(TFS_THREADS) raster_tf_scatter_kernel(
    int P /* view instances: views x Gaussians */, int Pv /* Gaussians per view */, int gy, uint32_t per_wg, uint32_t pthreads, int gx,
    uint32_t T /* LISTS = tiles (of all stacked views) x slabs */, const TFSlabs slabs,
    const uint32_t *__restrict__ rects, const float4 *__restrict__ rec,
    const uint32_t *__restrict__ depth_key, const uint32_t *__restrict__ tiles_touched, const uint32_t *__restrict__ wgoff,
    const uint32_t *__restrict__ wgmm, uint32_t producers, const TFCounters *__restrict__ ctr, uint32_t *__restrict__ words,
    uint32_t *__restrict__ mailbox, uint32_t seq, uint32_t cap, uint2 *__restrict__ pairs, WorkListOut wo,
    uint4 *__restrict__ big_parts, uint32_t big_cap, uint4 *__restrict__ small_tiles,
    uint32_t *__restrict__ nparts /* [2]: big parts, small tiles */)
{
    extern __shared__ uint32_t s_pos[];   // [T] start of the tile's segment + this workgroup's offset in it, bumped per instance
    __shared__ uint32_t s_wsum[3][TFS_THREADS / 64], s_carry[3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t *__restrict__ tile_count = ctr->tile_count;
    R2_TS_AT(tilefirst, 0);
    // the call's totals are where the preprocess kernel's atomics left them (round 4: its last workgroup copied them to the host
    // words in an epilogue that every later kernel waited for)
    const unsigned long long tot = ctr->total;
    const uint32_t R = tf_total_instances(tot);
    if (blockIdx.x == 0) {
        // ---- first of all, what the HOST is waiting for: {num_rendered, thin flag, key range, visible count} to the state's host
        // words and to the mailbox.  Key range of the call = max over the producers' slots
        uint32_t gkmax = 0u, gnkmin = 0u;
        for (uint32_t b = tid; b < producers; b += TFS_THREADS) {
            gkmax = max(gkmax, wgmm[2u * b]);
            gnkmin = max(gnkmin, wgmm[2u * b + 1u]);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            gkmax = max(gkmax, (uint32_t)__shfl_xor(gkmax, d));
            gnkmin = max(gnkmin, (uint32_t)__shfl_xor(gnkmin, d));
        }
        if (lane == 0) { s_wsum[0][wave] = gkmax; s_wsum[1][wave] = gnkmin; }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int q = 0; q < TFS_THREADS / 64; ++q) { gkmax = max(gkmax, s_wsum[0][q]); gnkmin = max(gnkmin, s_wsum[1][q]); }
            const uint32_t nvis = (uint32_t)(tot >> 40), any_thin = ctr->thin;
            words[DW_TOTAL] = R; words[DW_OVERFLOW] = 0u; words[DW_USER] = any_thin; words[DW_PMAX] = TF_MARK;
            words[DW_PNMAX] = 0u; words[DW_NMAX] = gkmax; words[DW_NNMAX] = gnkmin; words[DW_NVIS] = nvis;
            if (mailbox != nullptr) {   // (null on the second pass after a short prediction: the host has the totals already)
                mailbox[DW_TOTAL] = R; mailbox[DW_OVERFLOW] = 0u; mailbox[DW_USER] = any_thin; mailbox[DW_PMAX] = TF_MARK;
                mailbox[DW_PNMAX] = 0u; mailbox[DW_NMAX] = gkmax; mailbox[DW_NNMAX] = gnkmin; mailbox[DW_NVIS] = nvis;
                __hip_atomic_store(&mailbox[15], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        __syncthreads();
        R2_TS_AT(tilefirst, 5);
    }
    if (R > cap) {   // the buffers were sized by a prediction that fell short: do nothing, the host sizes them exactly and re-runs
        if (blockIdx.x == 0 && tid == 0) { wo.chunk_base[wo.T] = 0u; wo.chunk_base[wo.T + 1] = 0u; nparts[0] = 0u; nparts[1] = 0u; }
        return;
    }
    if (blockIdx.x == 0) return;
    if (blockIdx.x == 1) {
        // ---- tile ranges + the render kernel's work list (arrival counters zeroed, empty tiles appended); a tile's list is its
        // slabs' lists, one after the other
        if (SL) {
            for (uint32_t t = tid; t < wo.T; t += TFS_THREADS) {
                uint32_t c = 0u;
                for (uint32_t d = 0; d < slabs.n; ++d) c += tile_count[t * slabs.n + d];
                s_pos[t] = c;
            }
            __syncthreads();
        }
        ranges_and_work_block<TFS_THREADS>(SL ? s_pos : tile_count, wo);
        R2_TS_AT(tilefirst, 2);
        return;
    }
    if (blockIdx.x == 2) {
        // ---- the sort kernel's two lists
        if (tid < 3) s_carry[tid] = 0u;
        __syncthreads();
        for (uint32_t base = 0; base < T; base += TFS_THREADS) {
            const uint32_t t = base + (uint32_t)tid;
            const uint32_t c = t < T ? tile_count[t] : 0u;
            const uint32_t nb = c > TFK_SMALL_CAP ? (c > TFK_BIG_CAP ? (c + TFK_BIG_TARGET - 1u) / TFK_BIG_TARGET : 1u) : 0u;
            const uint32_t ns = (c != 0u && nb == 0u) ? 1u : 0u;
            uint32_t i0 = c, i1 = nb, i2 = ns;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t u0 = __shfl_up(i0, d), u1 = __shfl_up(i1, d), u2 = __shfl_up(i2, d);
                if (lane >= d) { i0 += u0; i1 += u1; i2 += u2; }
            }
            if (lane == 63) { s_wsum[0][wave] = i0; s_wsum[1][wave] = i1; s_wsum[2][wave] = i2; }
            __syncthreads();
            uint32_t o0 = 0, o1 = 0, o2 = 0;
            for (int w = 0; w < wave; ++w) { o0 += s_wsum[0][w]; o1 += s_wsum[1][w]; o2 += s_wsum[2][w]; }
            const uint32_t start = s_carry[0] + o0 + i0 - c, bstart = s_carry[1] + o1 + i1 - nb, sstart = s_carry[2] + o2 + i2 - ns;
            for (uint32_t q = 0; q < nb; ++q)
                if (bstart + q < big_cap) big_parts[bstart + q] = make_uint4(t, q | (nb << 16), start, c);
            if (ns) small_tiles[sstart] = make_uint4(t, 0u, start, c);
            __syncthreads();
            if (tid == TFS_THREADS - 1) { s_carry[0] = start + c; s_carry[1] = bstart + nb; s_carry[2] = sstart + ns; }
            __syncthreads();
        }
        if (tid == 0) { nparts[0] = min(s_carry[1], big_cap); nparts[1] = s_carry[2]; }
        R2_TS_AT(tilefirst, 2);
        return;
    }
    const uint32_t wg = blockIdx.x - TFS_SERVICE;
    // this thread's Gaussians (tf_grid: up to TF_PER_THREAD_MAX of the workgroup's range): requested now, used after the scan
    // (everything here was written by the previous kernel on other XCDs, i.e. comes from memory: one round trip for all of it
    // instead of one per dependent step)
    constexpr int NI = (int)TF_PER_THREAD_MAX;
    const uint32_t g0 = wg * per_wg, g1 = min(g0 + per_wg, (uint32_t)P);
    uint32_t g_idx[NI], g_tt[NI], g_rect[NI], g_key[NI];
    float4 g_box[NI];   // {px, py, hx, hy}: the bounding box of alpha >= 1e-5, for the instances' block masks (block_mask4)
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        g_idx[it] = g0 + (uint32_t)it * pthreads + (uint32_t)tid;
        const bool own = (uint32_t)tid < pthreads && g_idx[it] < g1;
        const uint32_t idc = min(g_idx[it], (uint32_t)P - 1u);
        g_tt[it] = tiles_touched[idc]; g_rect[it] = rects[idc]; g_key[it] = depth_key[idc];
        const float4 ra = rec[2u * idc], rb = rec[2u * idc + 1u];   // (culled Gaussians: stale words, never used -- g_tt is 0)
        g_box[it] = make_float4(ra.x, ra.y, rb.z, rb.w);
        g_tt[it] = own ? g_tt[it] : 0u;
    }
    // ---- exclusive scan of the tile counts (every workgroup for itself: <= 16 KB, cheaper than a launch boundary)
    const uint32_t *__restrict__ my_off = wgoff + (size_t)wg * T;
    constexpr int MAXT = (int)(TF_MAX_TILES / TFS_THREADS);
    uint32_t offs[MAXT];
#pragma unroll
    for (int q = 0; q < MAXT; ++q) {
        const uint32_t t = (uint32_t)(q * TFS_THREADS + tid);
        offs[q] = my_off[min(t, T - 1u)];
        s_pos[min(t, T - 1u)] = tile_count[min(t, T - 1u)];   // (clamped duplicates store the same value)
    }
    __syncthreads();
    const uint32_t ipt = (T + TFS_THREADS - 1) / TFS_THREADS;
    const uint32_t t0 = (uint32_t)tid * ipt, t1 = min(t0 + ipt, T);
    uint32_t sum = 0;
    for (uint32_t t = t0; t < t1; ++t) sum += s_pos[t];
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) s_wsum[0][wave] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (int w = 0; w < wave; ++w) run += s_wsum[0][w];
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t c = s_pos[t];
        s_pos[t] = run;
        run += c;
    }
    __syncthreads();
    // ... + where this workgroup's instances start inside each segment (rows of tiles it does not touch hold stale words:
    // they are added to slots nobody reads)
#pragma unroll
    for (int q = 0; q < MAXT; ++q) {
        const uint32_t t = (uint32_t)(q * TFS_THREADS + tid);
        if (t < T) s_pos[t] += offs[q];
    }
    __syncthreads();
    // ---- every instance of this workgroup's Gaussians
#pragma unroll
    for (int it = 0; it < NI; ++it)
        if (g_tt[it] != 0u) {
            const uint32_t rect = g_rect[it], key = g_key[it];
            const uint32_t x0 = rect & 0xFFu, y0 = (rect >> 8) & 0xFFu, w = ((rect >> 16) & 0xFFu) + 1u, h = (rect >> 24) + 1u;
            const uint32_t nsl = SL ? slabs.n : 1u, sl = SL ? tf_slab_of(key, slabs) : 0u;
            // second word: id << MASK_BITS | the instance's block mask -- sorting on (key, word) is sorting on (key, id).
            // block_mask4 without a float operation per instance (a thread walks its rectangle serially: what the inner loop costs
            // is what the workgroup's longest lane costs -- the four float tests per instance of the first version took the kernel
            // from 13.9 to 16.5 us, from 30 to 37 us on the 331k trained cloud with its 9 instances per Gaussian).  block_live is a
            // product of an x and a y test against INTEGER pixel bounds: px - hx <= X + 7 <=> ceil(px - hx) <= X + 7, and
            // px + hx >= X <=> floor(px + hx) >= X.  So the 8-pixel blocks the box touches, counted from the rectangle's first tile, are
            // the range [(A - X0) >> 3, (B - X0) >> 3] (A = ceil(px - hx), B = floor(px + hx), X0 = the rectangle's first pixel;
            // arithmetic shifts): one bit mask per axis and Gaussian, two bit-field extracts and a multiply per instance -- the same
            // masks, bit for bit.  Blocks beyond the 32nd of a rectangle (more than 16 tiles across) count as live: a mask may
            // name a block the box misses (the render kernels then evaluate pixels that all fail the cut-off), never miss one.
            const uint32_t idw = g_idx[it] << MASK_BITS;
            // (stacked views: the rectangle's rows count from the top of the stack, the record's pixel coordinates from its view's)
            const uint32_t yoff = Pv == P ? 0u : (g_idx[it] / (uint32_t)Pv) * (uint32_t)gy;
            auto axis_mask = [](float c, float hw, int origin) -> uint32_t {
                const float lim = 1048576.0f;   // (+-inf half-widths: never / always live)
                const int A = (int)fminf(fmaxf(ceilf(c - hw), -lim), lim), B = (int)fminf(fmaxf(floorf(c + hw), -lim), lim);
                const int lo = max((A - origin) >> 3, 0), hi = min((B - origin) >> 3, 31);
                return lo > hi ? 0u : ((0xFFFFFFFFu >> (31 - hi)) & (0xFFFFFFFFu << lo));
            };
            const uint32_t xm = axis_mask(g_box[it].x, g_box[it].z, (int)(x0 * (uint32_t)TILE2D));
            const uint32_t ym = axis_mask(g_box[it].y, g_box[it].w, (int)((y0 - yoff) * (uint32_t)TILE2D));
            for (uint32_t r = 0; r < h; ++r) {
                const uint32_t row = ((y0 + r) * (uint32_t)gx + x0) * nsl + sl;
                const uint32_t yb = r < 16u ? (ym >> (2u * r)) & 3u : 3u;
                const uint32_t ymul = (yb & 1u) | ((yb & 2u) << 1);   // y block 0 owns mask bits 0-1, y block 1 bits 2-3
                for (uint32_t c = 0; c < w; ++c) {
                    const uint32_t xb = c < 16u ? (xm >> (2u * c)) & 3u : 3u;
                    const uint32_t pos = atomicAdd(&s_pos[row + c * nsl], 1u);
                    pairs[pos] = make_uint2(key, idw | (xb * ymul));
                }
            }
        }
    R2_TS_AT(tilefirst, 1);
}

// ---- 3. per-tile sort
__device__ __forceinline__ unsigned long long tf_pack(const uint2 e) { return ((unsigned long long)e.x << 32) | (unsigned long long)e.y; }

// (kmin, kmax) over the NT threads of a group (waves w0 .. w0 + NT / 64) -> every thread of the group; two workgroup barriers
template <int NT>
__device__ __forceinline__ void tf_group_minmax(uint32_t &kmin, uint32_t &kmax, uint32_t (*s_mm)[TFK_THREADS / 64], int lane, int wave, int w0)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor(kmin, d));
        kmax = max(kmax, (uint32_t)__shfl_xor(kmax, d));
    }
    __syncthreads();
    if (lane == 0) { s_mm[0][wave] = kmin; s_mm[1][wave] = kmax; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { kmin = min(kmin, s_mm[0][w0 + w]); kmax = max(kmax, s_mm[1][w0 + w]); }
}

// mine[u] (entry u * NT + gtid of the group's cnt entries, in registers; ~0 beyond cnt) -> dst[rank by (key, id)] = id.
// One-level bucket sort (see voxel_small.hip for the reasoning: a tile's depth keys are float bit patterns in a narrow range, about
// one entry per value-linear bucket; position = bucket base + rank inside the bucket by (key, id), exact whatever the
// distribution).  A group of NT threads (gtid = thread inside it, waves w0 ..); s_a [>= cnt] receives the entries bucket by
// bucket; the barriers are the WORKGROUP's: every group of the workgroup calls this together.
template <int NT, uint32_t PER, uint32_t BINS>
__device__ __forceinline__ void tf_sort_group(unsigned long long (&mine)[PER], unsigned long long *s_a, uint32_t *s_bin,
                                              uint32_t *s_wsum /* workgroup's, per wave */, int gtid, int lane, int wave, int w0,
                                              uint32_t cnt, uint32_t kmin, uint32_t kmax, uint32_t *__restrict__ dst,
                                              uint32_t *__restrict__ dst_masked)
{
    R2_TS_AT(tilefirst, 6);
    for (uint32_t i = gtid; i <= BINS; i += NT) s_bin[i] = 0u;
    __syncthreads();
    const float scale = kmax > kmin ? (float)(BINS - 1) / (float)(kmax - kmin) : 0.f;
    uint32_t my_bin[PER], my_ticket[PER];
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
        const uint32_t i = u * NT + (uint32_t)gtid;
        my_bin[u] = 0u; my_ticket[u] = 0u;
        if (i < cnt) {
            my_bin[u] = min((uint32_t)((float)((uint32_t)(mine[u] >> 32) - kmin) * scale), BINS - 1u);
            my_ticket[u] = atomicAdd(&s_bin[my_bin[u]], 1u);
        }
    }
    __syncthreads();
    R2_TS_AT(tilefirst, 7);
    {
        constexpr uint32_t BPT = BINS / NT;
        static_assert(BINS % NT == 0, "whole buckets per thread");
        uint32_t c[BPT], tsum = 0;
#pragma unroll
        for (uint32_t q = 0; q < BPT; ++q) { c[q] = s_bin[gtid * BPT + q]; tsum += c[q]; }
        uint32_t incl = tsum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t run = incl - tsum;
        for (int w = w0; w < wave; ++w) run += s_wsum[w];
#pragma unroll
        for (uint32_t q = 0; q < BPT; ++q) { s_bin[gtid * BPT + q] = run; run += c[q]; }
        if (gtid == NT - 1) s_bin[BINS] = run;
    }
    __syncthreads();
    R2_TS_AT(tilefirst, 8);
    uint32_t b0[PER], b1[PER];
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
        const uint32_t i = u * NT + (uint32_t)gtid;
        b0[u] = 0u; b1[u] = 0u;
        if (i < cnt) {
            b0[u] = s_bin[my_bin[u]];
            b1[u] = s_bin[my_bin[u] + 1u];
            if (b1[u] - b0[u] > 1u) s_a[b0[u] + my_ticket[u]] = mine[u];   // a bucket of one needs neither the store nor a rank
        }
    }
    __syncthreads();
    R2_TS_AT(tilefirst, 9);
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
        const uint32_t i = u * NT + (uint32_t)gtid;
        if (i < cnt) {
            uint32_t r = 0;
            if (b1[u] - b0[u] > 1u)
                for (uint32_t q = b0[u]; q < b1[u]; ++q) r += s_a[q] < mine[u] ? 1u : 0u;
            b0[u] += r;   // the entry's position in the sorted list
        }
    }
    // the ids go to their positions in LDS first and leave in list order with unit-stride stores (round 5, from the voxelizer's
    // stick chain, where it took 10 us off a 66 us kernel: written straight from the lanes that hold them, a wave's store touches
    // 64 cache lines.  Here: sort stage 20.4 -> 19.8 us, the step unchanged -- this kernel is its round trips)
    uint32_t *s_out = reinterpret_cast<uint32_t *>(s_a);
    __syncthreads();   // the ranks have been read
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u)
        if (u * NT + (uint32_t)gtid < cnt) s_out[b0[u]] = (uint32_t)mine[u];
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
        const uint32_t j = u * NT + (uint32_t)gtid;
        if (j < cnt) {
            const uint32_t v = s_out[j];
            dst[j] = v >> MASK_BITS;   // point_list: the reference's, bit for bit
            dst_masked[j] = v;         // the render kernels' list: the same ids with their block masks
        }
    }
}

__global__ void __launch_bounds__(TFK_THREADS, 8) raster_tf_sort_kernel(
    const uint4 *__restrict__ big_parts, uint32_t big_cap, const uint4 *__restrict__ small_tiles, const uint32_t *__restrict__ nparts,
    const uint2 *__restrict__ pairs, uint32_t *__restrict__ point_list, TFCounters *__restrict__ ctr,
    const uint32_t *__restrict__ words, uint32_t cap)
{
    extern __shared__ unsigned long long tfk_lds[];
    __shared__ uint32_t s_mm[2][TFK_THREADS / 64], s_wsum[TFK_THREADS / 64], s_cnt, s_off;
    const uint32_t p = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *__restrict__ tile_count = ctr->tile_count;
    // the masked list sits behind point_list and the tile ids, where the TRUE instance count puts it (RasterBinning::carve)
    uint32_t *__restrict__ masked = binning_masked_ptr(reinterpret_cast<char *>(point_list), (size_t)words[DW_TOTAL]);
    R2_TS_AT(tilefirst, 3);
    // the scalar counters have been read for the last time by the scatter kernel: ready for the next call -- unless the state was
    // too small for this one, in which case the scatter kernel will want them again
    if (p == 0u && tid == 0 && words[DW_TOTAL] <= cap) {
        __hip_atomic_store(&ctr->total, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ctr->thin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the descriptor is requested together with the counts that say whether it exists (one round trip, not two): the big-part
    // list has a slot for every workgroup below its capacity; the short list is indexed behind the big count, so it waits
    const uint4 pd = big_parts[min(p, big_cap - 1u)];
    const uint32_t nbig = nparts[0], nsmall = nparts[1];
    if (p >= nbig) {
        // ---- group: TFK_GROUPS short lists, one per 256 threads of the workgroup
        const uint32_t q0 = (p - nbig) * (uint32_t)TFK_GROUPS;
        if (q0 >= nsmall) return;
        const int g = tid >> 8, gtid = tid & 255, w0 = g * 4;
        unsigned long long *s_a = tfk_lds + (size_t)g * TFK_SMALL_CAP;
        uint32_t *s_bin = reinterpret_cast<uint32_t *>(tfk_lds + TFK_GROUPS * TFK_SMALL_CAP) + (size_t)g * (TFK_SMALL_BINS + 1);
        uint32_t n = 0, start = 0;
        if (q0 + (uint32_t)g < nsmall) {
            const uint4 sd = small_tiles[q0 + g];
            start = sd.z; n = sd.w;
            if (gtid == 0) tile_count[sd.x] = 0u;   // this call has read it for the last time: ready for the next one
        }
        uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
        unsigned long long mine[TFK_SMALL_PER];
        {
            // all loads in flight before the first use, BRANCH-FREE (clamped addresses): behind a per-lane branch hipcc drains the
            // load queue at every use (s_waitcnt vmcnt(0)) -- six dependent round trips instead of one
            uint2 e[TFK_SMALL_PER];
            const uint32_t last = start + (n ? n - 1u : 0u);
#pragma unroll
            for (uint32_t u = 0; u < TFK_SMALL_PER; ++u) e[u] = pairs[min(start + u * 256u + (uint32_t)gtid, last)];
#pragma unroll
            for (uint32_t u = 0; u < TFK_SMALL_PER; ++u) {
                const bool in = u * 256u + (uint32_t)gtid < n;
                mine[u] = in ? tf_pack(e[u]) : ~0ull;
                kmin = in ? min(kmin, e[u].x) : kmin;
                kmax = in ? max(kmax, e[u].x) : kmax;
            }
        }
        tf_group_minmax<256>(kmin, kmax, s_mm, lane, wave, w0);
        tf_sort_group<256, TFK_SMALL_PER, TFK_SMALL_BINS>(mine, s_a, s_bin, s_wsum, gtid, lane, wave, w0, n, kmin, kmax, point_list + start,
                                                          masked + start);
        R2_TS_AT(tilefirst, 4);
        return;
    }
    // ---- big: one list (or one depth range of a very long one) for the whole workgroup
    unsigned long long *s_a = tfk_lds;                                             // [TFK_BIG_CAP]
    uint32_t *s_bin = reinterpret_cast<uint32_t *>(s_a + TFK_BIG_CAP);             // [TFK_BIG_BINS + 1]
    uint32_t *s_coarse = s_bin + TFK_BIG_BINS + 1;                                 // [TFK_COARSE + 1]
    const uint32_t tile = pd.x, part = pd.y & 0xFFFFu, nparts_tile = pd.y >> 16, start = pd.z, n = pd.w;
    if (part == 0u && tid == 0) tile_count[tile] = 0u;
    const uint2 *__restrict__ src = pairs + start;
    uint32_t *__restrict__ dst = point_list + start;
    uint32_t *__restrict__ dst_masked = masked + start;
    uint32_t cnt = n, out_off = 0u;
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    unsigned long long mine[TFK_BIG_PER];
    if (nparts_tile == 1u) {
        uint2 e[TFK_BIG_PER];   // all loads in flight before the first use, branch-free (see the quad path)
#pragma unroll
        for (uint32_t u = 0; u < TFK_BIG_PER; ++u) e[u] = src[min(u * TFK_THREADS + (uint32_t)tid, n - 1u)];
#pragma unroll
        for (uint32_t u = 0; u < TFK_BIG_PER; ++u) {
            const bool in = u * TFK_THREADS + (uint32_t)tid < n;
            mine[u] = in ? tf_pack(e[u]) : ~0ull;
            kmin = in ? min(kmin, e[u].x) : kmin;
            kmax = in ? max(kmax, e[u].x) : kmax;
        }
        tf_group_minmax<TFK_THREADS>(kmin, kmax, s_mm, lane, wave, 0);
    } else {
        // ---- this part's share of a very long list: a contiguous range of the bins of a depth histogram of the list (bins laid
        // over the key range of the call, which the preprocess left in the host words), chosen so that the parts are balanced
        // whatever the distribution.  The histogram is built from every 8th entry -- every part of the tile takes the same
        // sample, hence the same split; the split only balances the parts, membership and offsets below are exact.
        constexpr uint32_t SAMPLE = 8;
        const uint32_t ns = (n + SAMPLE - 1u) / SAMPLE;
        const uint32_t gmax = words[DW_NMAX];
        uint32_t lo = ~words[DW_NNMAX];
        float cscale = gmax > lo ? (float)(TFK_COARSE - 1) / (float)(gmax - lo) : 0.f;
        auto coarse_of = [&](uint32_t kk) { return min((uint32_t)((float)(max(kk, lo) - lo) * cscale), TFK_COARSE - 1u); };
        auto sample_hist = [&](bool track) {
            for (uint32_t base = 0; base < ns; base += 4u * TFK_THREADS) {   // four loads in flight, branch-free
                uint32_t kk[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) kk[u] = src[min((base + u * TFK_THREADS + (uint32_t)tid) * SAMPLE, n - 1u)].x;
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u)
                    if (base + u * TFK_THREADS + (uint32_t)tid < ns) {
                        atomicAdd(&s_coarse[coarse_of(kk[u])], 1u);
                        if (track) { kmin = min(kmin, kk[u]); kmax = max(kmax, kk[u]); }
                    }
            }
        };
        for (uint32_t i = tid; i <= TFK_COARSE; i += TFK_THREADS) s_coarse[i] = 0u;
        if (tid == 0) { s_cnt = 0u; s_off = 0u; }
        __syncthreads();
        sample_hist(true);
        tf_group_minmax<TFK_THREADS>(kmin, kmax, s_mm, lane, wave, 0);   // (also orders the histogram's atomics before its readers)
        {
            // a list whose depths crowd into a few of those bins (a thin slab seen face-on) cannot be split there: lay the bins
            // over the sample's own key range instead (keys outside it clamp into the end bins)
            uint32_t mb = s_coarse[tid];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) mb = max(mb, (uint32_t)__shfl_xor(mb, d));
            __syncthreads();
            if (lane == 0) s_wsum[wave] = mb;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < TFK_THREADS / 64; ++w) mb = max(mb, s_wsum[w]);
            if (mb > TFK_BIG_CAP / (2u * SAMPLE) && (kmin > lo || kmax < gmax)) {   // workgroup-uniform
                __syncthreads();
                for (uint32_t i = tid; i <= TFK_COARSE; i += TFK_THREADS) s_coarse[i] = 0u;
                lo = kmin;
                cscale = kmax > kmin ? (float)(TFK_COARSE - 1) / (float)(kmax - kmin) : 0.f;
                __syncthreads();
                sample_hist(false);
            }
            __syncthreads();
        }
        kmin = 0xFFFFFFFFu; kmax = 0u;   // reused below for this part's own range
        // exclusive prefix over the bins (one per thread); owner of a bin = floor(prefix * parts / samples): monotone in the bin
        const uint32_t c = s_coarse[tid];
        uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t excl = incl - c;
        for (int w = 0; w < wave; ++w) excl += s_wsum[w];
        const uint32_t owner = min((uint32_t)(((unsigned long long)excl * nparts_tile) / ns), nparts_tile - 1u);
        __syncthreads();
        s_coarse[tid] = owner;   // from here on: the part that owns the bin
        __syncthreads();
        // ---- one pass over the whole list: entries of bins below mine are counted (my offset in the sorted list), mine appended
        uint32_t below = 0;
        bool overflow = false;
        for (uint32_t base4 = 0; base4 < n; base4 += 4u * TFK_THREADS) {   // whole waves stay in the loop: the append is wave-cooperative
            uint2 e4[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) e4[u] = src[min(base4 + u * TFK_THREADS + (uint32_t)tid, n - 1u)];
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
                const uint32_t i = base4 + u * TFK_THREADS + (uint32_t)tid;
                const uint2 e = e4[u];
                const uint32_t own = i < n ? s_coarse[coarse_of(e.x)] : 0xFFFFFFFFu;
                below += own < part ? 1u : 0u;
                const bool take = own == part;
                const unsigned long long mm = __ballot(take);
                if (mm) {
                    uint32_t wbase = 0;
                    const int leader = __ffsll((long long)mm) - 1;
                    if (lane == leader) wbase = atomicAdd(&s_cnt, (uint32_t)__popcll(mm));
                    wbase = __shfl(wbase, leader);
                    const uint32_t slot = wbase + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
                    if (take && slot < TFK_BIG_CAP) {
                        s_a[slot] = tf_pack(e);
                        kmin = min(kmin, e.x);
                        kmax = max(kmax, e.x);
                    }
                }
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) below += __shfl_xor(below, d);
        if (lane == 0) atomicAdd(&s_off, below);
        __syncthreads();
        cnt = s_cnt;
        out_off = s_off;
        overflow = cnt > TFK_BIG_CAP;
        if (cnt == 0u) return;
        if (overflow) {
            // (more entries of one tile inside a sliver of its depth range than the LDS holds: equal depths.)  Rank by counting,
            // straight from memory: O(cnt x n).  Exact like everything else.
            for (uint32_t i = tid; i < n; i += TFK_THREADS) {
                const uint2 e = src[i];
                if (s_coarse[coarse_of(e.x)] != part) continue;
                const unsigned long long me = tf_pack(e);
                uint32_t r = 0;
                for (uint32_t j = 0; j < n; ++j) {
                    const uint2 o = src[j];
                    if (s_coarse[coarse_of(o.x)] == part && tf_pack(o) < me) ++r;
                }
                dst[out_off + r] = e.y >> MASK_BITS;
                dst_masked[out_off + r] = e.y;
            }
            return;
        }
        tf_group_minmax<TFK_THREADS>(kmin, kmax, s_mm, lane, wave, 0);   // (its barriers also order the appends before the reads)
#pragma unroll
        for (uint32_t u = 0; u < TFK_BIG_PER; ++u) {
            const uint32_t i = u * TFK_THREADS + (uint32_t)tid;
            mine[u] = i < cnt ? s_a[i] : ~0ull;
        }
        __syncthreads();   // everybody holds its entries: s_a may be overwritten by the placement
    }
    tf_sort_group<TFK_THREADS, TFK_BIG_PER, TFK_BIG_BINS>(mine, s_a, s_bin, s_wsum, tid, lane, wave, 0, cnt, kmin, kmax, dst + out_off,
                                                          dst_masked + out_off);
    R2_TS_AT(tilefirst, 4);
}

// ---- host side
// Counters that several workgroups bump with atomics live in a small persistent allocation per (host thread, device, stream):
// self-resetting, so no zero-fill launch precedes the forward.  Owned by the thread: freed when it exits (or on
// r2_thread_release()); a thread that cycles through more streams than the table holds evicts the least recently used entry
// (round 4: never freed, and the fast path silently switched itself off after 64 streams -- ADVICE r4).
struct TFWorkspace { int dev; hipStream_t stream; TFCounters *ctr; uint32_t *nparts; bool dirty; unsigned long long used; int epoch; };
// a deferred forward whose prediction fell short leaves its counters behind (nobody re-runs the chain): every workspace is then
// cleaned before its next use (the backward that finds out runs on another thread: a process-wide epoch)
std::atomic<int> g_tf_clean_epoch{0};
struct TFWorkspaces {
    std::vector<TFWorkspace> v;
    unsigned long long tick = 0;
    static void free_one(const TFWorkspace &w)
    {
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return; }
        if (cur != w.dev && hipSetDevice(w.dev) != hipSuccess) { (void)hipGetLastError(); return; }
        if (hipFree(w.ctr) != hipSuccess) (void)hipGetLastError();   // (its last forward may still run: hipFree waits for the device)
        if (cur != w.dev) (void)hipSetDevice(cur);
    }
    void release()
    {
        for (const TFWorkspace &w : v) free_one(w);
        v.clear();
    }
    ~TFWorkspaces() { release(); }
};
thread_local TFWorkspaces g_tf_ws;
constexpr size_t TF_MAX_WORKSPACES = 16;

TFWorkspace *tf_workspace(int dev, hipStream_t s)
{
    TFWorkspaces &t = g_tf_ws;
    ++t.tick;
    for (TFWorkspace &w : t.v)
        if (w.dev == dev && w.stream == s) { w.used = t.tick; return &w; }
    if (t.v.size() >= TF_MAX_WORKSPACES) {
        size_t lru = 0;
        for (size_t i = 1; i < t.v.size(); ++i)
            if (t.v[i].used < t.v[lru].used) lru = i;
        TFWorkspaces::free_one(t.v[lru]);
        t.v.erase(t.v.begin() + (long)lru);
    }
    char *p = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&p), sizeof(TFCounters) + 64) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    t.v.push_back(TFWorkspace{dev, s, reinterpret_cast<TFCounters *>(p), reinterpret_cast<uint32_t *>(p + sizeof(TFCounters)), true, t.tick,
                              g_tf_clean_epoch.load(std::memory_order_relaxed)});
    return &t.v.back();
}

// the thread's recent instance counts per problem size: what the prediction is made of
struct TFHint {
    int P, V, W, H;   // Gaussians, stacked views, detector
    uint32_t recent[8], n;
    bool thin;
    uint32_t kmax, kmin;    // depth-key range of the last call (the next call's slabs are laid over it)
    unsigned long long used;
};
thread_local std::vector<TFHint> g_tf_hints;
thread_local unsigned long long g_tf_hint_tick = 0;

TFHint *tf_hint(int P, int V, int W, int H, bool create)
{
    for (TFHint &h : g_tf_hints)
        if (h.P == P && h.V == V && h.W == W && h.H == H) { h.used = ++g_tf_hint_tick; return &h; }
    if (!create) return nullptr;
    if (g_tf_hints.size() >= 16) {
        size_t lru = 0;
        for (size_t i = 1; i < g_tf_hints.size(); ++i)
            if (g_tf_hints[i].used < g_tf_hints[lru].used) lru = i;
        g_tf_hints.erase(g_tf_hints.begin() + (long)lru);
    }
    g_tf_hints.push_back(TFHint{P, V, W, H, {0}, 0u, false, 0u, 0u, ++g_tf_hint_tick});
    return &g_tf_hints.back();
}

// no call with this P yet -- every densification changes it (train.py:155-168) --: the thread's most recent call on the same
// detector, scaled by the ratio of the Gaussian counts
const TFHint *tf_hint_nearby(int V, int W, int H)
{
    const TFHint *best = nullptr;
    for (const TFHint &h : g_tf_hints)
        if (h.V == V && h.W == W && h.H == H && h.n != 0u && h.P > 0 && (!best || h.used > best->used)) best = &h;
    return best;
}

// what the chain did, process-wide (r2_tile_first_stats): forwards that took it, forwards it declined (no prediction / limits),
// chains run a second time because the prediction fell short, renders repeated for the thin-Gaussian variant, forwards whose
// prediction was seeded from another P (the call after a densification)
std::atomic<long long> g_tf_taken{0}, g_tf_declined{0}, g_tf_rerun{0}, g_tf_rerender{0}, g_tf_seeded{0};

std::atomic<int> g_tf_mode{-1};   // -1: not decided yet (environment), 0: off, 1: on
std::atomic<int> g_defer_mode{-1};   // deferred num_rendered (r2_defer_count_control): -1 environment, 0 off (default), 1 on
std::atomic<long long> g_defer_taken{0}, g_defer_no_slot{0}, g_defer_short{0};

bool defer_enabled()
{
    int on = g_defer_mode.load(std::memory_order_relaxed);
    if (on < 0) {
        const char *e = getenv("R2_DEFER_COUNT");
        on = (e && e[0] == '1') ? 1 : 0;
        g_defer_mode.store(on, std::memory_order_relaxed);
    }
    return on != 0;
}

// deferred forwards of this thread whose instance count it has not seen yet: polled (non-blocking) at its next forward, so that its
// predictions stay current although the count is resolved by the backward's thread
struct TFPending { int token, P, V, W, H; };
thread_local std::vector<TFPending> g_tf_pending;

bool tf_switched_on()
{
    int on = g_tf_mode.load(std::memory_order_relaxed);
    if (on < 0) {
        const char *e = getenv("R2_TILE_FIRST");
        on = (e && e[0] == '0') ? 0 : 1;
        g_tf_mode.store(on, std::memory_order_relaxed);
    }
    return on != 0;
}
bool tf_lds_ok()
{
    static signed char lds_state[R2_MAX_DEVICES] = {};
    return TFK_LDS <= device_lds_optin_bytes() &&
           allow_dynamic_lds(reinterpret_cast<const void *>(raster_tf_sort_kernel), (int)TFK_LDS, lds_state);
}

int tf_forced_slabs()
{
    static const int v = [] { const char *e = getenv("R2_TF_SLABS"); return e ? atoi(e) : 0; }();
    return (v == 1 || v == 2 || v == 4) ? v : 0;
}

}  // namespace

void raster_tilefirst_note(int P, int V, int W, int H, uint32_t num_rendered, bool thin, uint32_t kmax, uint32_t kmin)
{
    TFHint *h = tf_hint(P, V, W, H, true);
    h->recent[h->n++ & 7u] = num_rendered;
    h->thin = thin;
    h->kmax = kmax;
    h->kmin = kmin;
}

// -> num_rendered (>= 0), a negative error code, or TF_NOT_TAKEN: nothing was launched, run the general path
int raster_forward_tilefirst(const char *what, r2_alloc_fn geometryBuffer, void *geometry_user, r2_alloc_fn binningBuffer,
                             void *binning_user, r2_alloc_fn imageBuffer, void *image_user, int P, int V, int width, int height,
                             const float *means3D, const float *opacities, const float *scales, float scale_modifier,
                             const float *rotations, const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix,
                             float tan_fovx, float tan_fovy, int mode, float *out_color, int *radii, hipStream_t s)
{
    // V stacked views (round 6; r2_raster_forward_batch): the chain runs over the V * P view instances on the stacked tile grid
    // (tile = (v * gy + ty) * gx + tx, raster_state.hpp) -- the three binning kernels are one round of workgroups each whatever the
    // work, so a batch pays for them once
    const int gx = (width + TILE2D - 1) / TILE2D, gy = (height + TILE2D - 1) / TILE2D;
    const size_t T = (size_t)gx * gy * (size_t)V, N = (size_t)width * height * (size_t)V;
    const size_t PVs = (size_t)P * (size_t)V;
    const TFGrid grid = tf_grid((int)std::min<size_t>(PVs, (size_t)1 << 24), device_cu_count());
    const size_t wgs = grid.wgs;
    // the static rule: dispatch.hpp (debug calls never get here)
    const bool on = tf_switched_on();
    const RasterChoice choice = raster_forward_choice((size_t)P, (size_t)V, width, height, false, on, on && tf_lds_ok(), wgs);
    if (!choice.tile_first) {
        path_count(choice.why);
        g_tf_declined.fetch_add(1, std::memory_order_relaxed);
        return TF_NOT_TAKEN;
    }
    const int PV = (int)PVs;
    for (size_t i = 0; i < g_tf_pending.size();) {
        uint32_t pw[DW_COUNT];
        const TFPending pn = g_tf_pending[i];
        if (defer_peek(pn.token, pw, DW_COUNT)) {
            raster_tilefirst_note(pn.P, pn.V, pn.W, pn.H, pw[DW_TOTAL], pw[DW_USER] != 0u, pw[DW_NMAX], ~pw[DW_NNMAX]);
            g_tf_pending.erase(g_tf_pending.begin() + (long)i);
        } else if (g_tf_pending.size() > 16) {
            g_tf_pending.erase(g_tf_pending.begin() + (long)i);   // (its slot was recycled, or the device is far behind)
        } else {
            ++i;
        }
    }
    // ---- the prediction: the largest count of the thread's recent calls with this P and detector; for a P it has not rendered
    // yet (the call after a densification) the most recent call on the same detector, scaled -- same scene, more Gaussians
    uint32_t rmax = 0, kmax = 0, kmin = 0;
    if (const TFHint *hint = tf_hint(P, V, width, height, false); hint && hint->n != 0u) {
        for (uint32_t i = 0; i < std::min(hint->n, 8u); ++i) rmax = std::max(rmax, hint->recent[i]);
        kmax = hint->kmax; kmin = hint->kmin;
    } else if (const TFHint *near = tf_hint_nearby(V, width, height)) {
        uint32_t r0 = 0;
        for (uint32_t i = 0; i < std::min(near->n, 8u); ++i) r0 = std::max(r0, near->recent[i]);
        const double f = (double)P / (double)near->P;
        rmax = (uint32_t)std::min<double>((double)r0 * f * 1.1, 2.0e9);
        kmax = near->kmax; kmin = near->kmin;
        g_tf_seeded.fetch_add(1, std::memory_order_relaxed);
        path_count(PS_RAS_EVENT_SEEDED);
    } else {
        path_count(PS_RAS_GENERAL_NO_PREDICTION);
        g_tf_declined.fetch_add(1, std::memory_order_relaxed);
        return TF_NOT_TAKEN;   // no prediction yet: the general path, which leaves one behind
    }
    // (deferred forwards of this thread that have completed meanwhile: their counts into the history -- done above the prediction)
    int dev = 0;
    R2_HIP_TRY(hipGetDevice(&dev));
    TFWorkspace *ws = tf_workspace(dev, s);
    if (!ws) {
        path_count(PS_RAS_GENERAL_NO_WORKSPACE);
        g_tf_declined.fetch_add(1, std::memory_order_relaxed);
        return TF_NOT_TAKEN;
    }
    path_count(PS_RAS_TILE_FIRST);
    // + 25 %, in steps of 64 K instances (the allocator behind the callbacks then sees few distinct sizes); a deferred forward,
    // which cannot repeat itself when the prediction falls short, takes + 50 %
    const bool defer = defer_enabled();
    size_t cap = (((size_t)rmax + (defer ? rmax / 2 : rmax / 4) + 16384) + 65535) & ~(size_t)65535;

    // depth slabs per tile, from an ESTIMATE of the longest tile list: a dense tile holds ~8x the mean on every scene seen so far
    // (synthetic 300k cloud: mean 1128, longest 8776; 92k trained cloud: 825 / 5856; 331k trained cloud: 3015 / 22388).  A wrong estimate costs
    // speed only.  (The exact length reported back by the scatter kernel through the mailbox was built and measured: +0.7 us on
    // that kernel for a decision the estimate gets right.)
    TFSlabs slabs{1u, kmin, 0.f};
    {
        const double ml = 8.0 * (double)rmax / (double)T;
        uint32_t d = 1u;
        if (ml > (double)TF_SLAB_SPLIT_ABOVE)
            while (d < TF_MAX_SLABS && T * d * 2u <= TF_MAX_TILES) d *= 2u;
        if (const int f = tf_forced_slabs())
            if (T * (size_t)f <= TF_MAX_TILES) d = (uint32_t)f;
        if (kmax <= kmin) d = 1u;   // no key range to lay the slabs over
        slabs.n = d;
        slabs.scale = d > 1u ? (float)d / ((float)(kmax - kmin) + 1.0f) : 0.f;
        if (d > 1u) path_count(PS_RAS_EVENT_DEPTH_SLABS);
    }
    const size_t TL = T * slabs.n;   // lists

    char *gchunk = geometryBuffer(RasterGeom::carve(nullptr, PV, TL, wgs).bytes, geometry_user);
    if (!gchunk) {
        set_error("%s: state allocation callback returned NULL", what);
        return R2_ERR_ALLOC;
    }
    const RasterGeom geom = RasterGeom::carve(gchunk, PV, TL, wgs);
    const int epoch = g_tf_clean_epoch.load(std::memory_order_relaxed);
    if (ws->dirty || ws->epoch != epoch)   // first use, a call that failed half way, or a deferred forward that fell short
        R2_HIP_TRY(hipMemsetAsync(ws->ctr, 0, sizeof(TFCounters) + 64, s));
    ws->dirty = true;
    ws->epoch = epoch;
    uint32_t *mailbox = nullptr, mailbox_seq = 0;
    int rc = 0, token = -1;
    if (defer && cap < (size_t)DEFER_TOKEN_FLAG) {
        token = defer_acquire(&mailbox, &mailbox_seq, (uint32_t)cap);
        if (token < 0) g_defer_no_slot.fetch_add(1, std::memory_order_relaxed);
    }
    if (token < 0) rc = host_mailbox_arm(&mailbox, &mailbox_seq);
    if (rc) return rc;
    { StageScope t(ST_RAS_PREPROCESS, s);
    launch_raster_preprocess_tf(geom, P, V, grid, slabs, means3D, scales, scale_modifier, rotations, opacities, cov3D_precomp, viewmatrix,
                                projmatrix, width, height, tan_fovx, tan_fovy, mode, radii, ws->ctr, s); }
    R2_HIP_TRY(hipGetLastError());

    RasterBinning bin{};
    RasterImage img{};
    // (post_mailbox: only the FIRST pass posts the totals to the host's mailbox -- by the second one the host has consumed them, and
    // a late post could land in a mailbox the thread has meanwhile armed for its next forward on another stream: ADVICE r5)
    bool post_mailbox = true;
    auto enqueue = [&](size_t capacity, bool render_only) -> int {
        if (!render_only) {
            char *bchunk = binningBuffer(RasterBinning::carve(nullptr, capacity).bytes, binning_user);
            char *ichunk = imageBuffer(RasterImage::carve(nullptr, T, N, capacity, false, TL).bytes, image_user);
            if (!bchunk || !ichunk) {
                set_error("%s: binning/image allocation callback returned NULL", what);
                return R2_ERR_ALLOC;
            }
            bin = RasterBinning::carve(bchunk, capacity);
            img = RasterImage::carve(ichunk, T, N, capacity, false, TL);
            uint2 *pairs = reinterpret_cast<uint2 *>(bin.part);   // backward scratch (32 bytes per instance), free until then
            { StageScope t(ST_RAS_DUPLICATE, s);
            const WorkListOut wo{img.ranges, img.chunk_base, img.work_tile, (uint32_t)T, FWD_CHUNK, img.tile_done, 0u, (uint32_t)img.NW,
                                 raster_forward_wave_kernel_on() ? 1u : 0u};
#define R2_TF_SCATTER(SLB)                                                                                                        \
            raster_tf_scatter_kernel<SLB><<<dim3((unsigned)wgs + TFS_SERVICE), dim3(TFS_THREADS), TL * sizeof(uint32_t), s>>>(         \
                PV, P, gy, grid.per_wg, grid.threads, gx, (uint32_t)TL, slabs, geom.tf_rect, geom.rec, geom.depth_key, geom.tiles_touched, geom.tf_wgoff, \
                geom.tf_wgmm, (uint32_t)wgs, ws->ctr, geom.host_words, post_mailbox ? mailbox : nullptr, mailbox_seq,                   \
                (uint32_t)std::min<size_t>(capacity, 0x7FFFFFFFu), pairs, wo, img.tf_parts, (uint32_t)img.NP, img.tf_parts + img.NP,   \
                ws->nparts)
            if (slabs.n > 1u) R2_TF_SCATTER(true);
            else R2_TF_SCATTER(false);
#undef R2_TF_SCATTER
            }
            R2_HIP_TRY(hipGetLastError());
            { StageScope t(ST_RAS_SORT, s);
            raster_tf_sort_kernel<<<dim3((unsigned)(img.NP + (TL + TFK_GROUPS - 1) / TFK_GROUPS)), dim3(TFK_THREADS), TFK_LDS, s>>>(
                img.tf_parts, (uint32_t)img.NP, img.tf_parts + img.NP, ws->nparts, pairs, bin.point_list, ws->ctr, geom.host_words,
                (uint32_t)std::min<size_t>(capacity, 0x7FFFFFFFu)); }
            R2_HIP_TRY(hipGetLastError());
        } else {
            R2_HIP_TRY(hipMemsetAsync(img.tile_done, 0, 4 * T * sizeof(uint32_t), s));   // the first render left its arrivals behind
        }
        { StageScope t(ST_RAS_RENDER_FWD, s);
        launch_raster_render_forward(geom, bin, img, width, height, V, out_color, false, bin.tiles, true, s,
                                     reinterpret_cast<char *>(bin.point_list), geom.host_words, (size_t)PV); }
        R2_HIP_TRY(hipGetLastError());
        return 0;
    };
    // (one render variant since round 6 -- thin Gaussians take its exact path, raster_render.hip: fwd_item -- so no render is ever repeated)
    rc = enqueue(cap, false);
    if (rc) return rc;
    if (token >= 0) {
        // ---- deferred: no wait.  The token goes back as num_rendered; the backward resolves it (raster_backward_impl)
        ws->dirty = false;   // (a count beyond the capacity is found by the backward, which bumps the clean epoch)
        g_tf_taken.fetch_add(1, std::memory_order_relaxed);
        g_defer_taken.fetch_add(1, std::memory_order_relaxed);
        path_count(PS_RAS_EVENT_DEFERRED);
        g_tf_pending.push_back(TFPending{token, P, V, width, height});
        host_mark_wait_begin();   // (r2_profile_host: a wait of zero length)
        host_mark_wait_end();
        return token;
    }
    uint32_t hw[DW_COUNT] = { 0 };
    rc = host_mailbox_wait(mailbox_seq, hw, DW_COUNT, s);
    if (rc) return rc;
    const uint32_t num_rendered = hw[DW_TOTAL];
    const bool thin = hw[DW_USER] != 0u;
    if (num_rendered > 0x7FFFFFFFu) {
        set_error("%s: more than 2147483647 (tile, Gaussian) instances: they do not fit the 31-bit num_rendered", what);
        return R2_ERR_INVALID;
    }
    post_mailbox = false;
    if ((size_t)num_rendered > cap) {
        g_tf_rerun.fetch_add(1, std::memory_order_relaxed);
        path_count(PS_RAS_EVENT_SECOND_PASS);
        rc = enqueue(num_rendered, false);      // the prediction fell short: exact sizes, same kernels
        if (rc) return rc;
    }
    ws->dirty = false;
    g_tf_taken.fetch_add(1, std::memory_order_relaxed);
    raster_tilefirst_note(P, V, width, height, num_rendered, thin, hw[DW_NMAX], ~hw[DW_NNMAX]);
    return (int)num_rendered;
}

void raster_tilefirst_release() { g_tf_ws.release(); g_tf_hints.clear(); g_tf_pending.clear(); }

// the backward's side of a deferred forward: token -> the true instance count (waits for the control words, which the forward's
// second kernel posted long ago); a count beyond the capacity the forward was launched with is an ERROR -- its kernels did nothing
int raster_resolve_deferred(const char *what, int token, hipStream_t s, uint32_t *num_rendered)
{
    uint32_t hw[DW_COUNT] = { 0 }, cap = 0;
    const int rc = defer_resolve(token, hw, DW_COUNT, &cap, s, true);
    if (rc) return rc;
    if (hw[DW_TOTAL] > cap) {
        g_defer_short.fetch_add(1, std::memory_order_relaxed);
        g_tf_clean_epoch.fetch_add(1, std::memory_order_relaxed);
        set_error("%s: the deferred forward's state was sized for %u instances, the scene has %u: its image and state are invalid "
                  "(R2_DEFER_COUNT / r2_defer_count_control trade the exact second pass for the missing wait; render this view again "
                  "with the mode off)", what, cap, hw[DW_TOTAL]);
        return R2_ERR_INVALID;
    }
    *num_rendered = hw[DW_TOTAL];
    return 0;
}

}  // namespace r2

extern "C" void r2_tile_first_stats(long long *out, int reset)
{
    std::atomic<long long> *c[5] = {&r2::g_tf_taken, &r2::g_tf_declined, &r2::g_tf_rerun, &r2::g_tf_rerender, &r2::g_tf_seeded};
    for (int i = 0; i < 5; ++i) {
        if (out) out[i] = c[i]->load(std::memory_order_relaxed);
        if (reset) c[i]->store(0, std::memory_order_relaxed);
    }
}

extern "C" void r2_defer_count_control(int mode)
{
    if (mode == 0 || mode == 1) r2::g_defer_mode.store(mode, std::memory_order_relaxed);
}

extern "C" void r2_defer_count_stats(long long *out, int reset)
{
    std::atomic<long long> *c[3] = {&r2::g_defer_taken, &r2::g_defer_no_slot, &r2::g_defer_short};
    for (int i = 0; i < 3; ++i) {
        if (out) out[i] = c[i]->load(std::memory_order_relaxed);
        if (reset) c[i]->store(0, std::memory_order_relaxed);
    }
}

extern "C" void r2_tile_first_control(int mode)
{
    if (mode == 0 || mode == 1) r2::g_tf_mode.store(mode, std::memory_order_relaxed);
    if (mode == 2) r2::g_tf_hints.clear();   // the calling thread's predictions
}
