// voxel_sticks.hip -- stick-first binning of the voxelizer (round 5): the reference's duplicateWithKeys -> SortPairs(tile | z bits) ->
// identifyTileRanges (VOX/voxelizer_impl.cu:54-128,244-287) for grids of 65 to 32 768 tiles (the 256^3 query of test.py:105-112 has
// 32 768; smaller grids take voxel_small.hip) without a global sort.
//
// Why.  The general chain (voxel_api.hip) orders the Gaussians by their z bits (five launches), emits the instances in that order
// and sorts them by their 15-bit tile id with two 8-bit radix passes (six launches) before the ranges and the work list (three):
// seventeen launches and ~185 us around a 343 us render at 300k Gaussians / 256^3 (profiles/r05f_voxel256_summary.md).  The
// rasterizer's tile-first chain (raster_tilefirst.hip) does not carry over as it is: 32 768 tile counters do not fit one LDS
// histogram, and one returning global atomic per (workgroup, tile) is 10^6 atomics here (the Gaussians of a workgroup are spread
// over the whole volume).  So:
//   * a LIST is a STICK of 2^shift consecutive tile ids (8 tiles along x at 256^3, one tile up to 4096 tiles): <= 4096 lists, one
//     LDS histogram;
//   * the per-(workgroup, list) counts go to memory with plain stores (16 KB per workgroup of 1024 Gaussians) and a column scan
//     turns them into offsets -- no global atomics except one per workgroup for the call's totals;
//   * every instance is then written straight into its list's segment as (z bits, id | tile-in-stick << 29), in arbitrary order;
//   * one workgroup (or a 256-thread quarter of one) per list sorts its segment in LDS by (tile-in-stick, z bits, id) -- the
//     bucket sort of raster_tilefirst.hip with the buckets divided among the stick's tiles -- and writes point_list, the
//     per-instance tile ids and the ranges of its tiles.
// Sorting every list by (tile, z bits, id) IS the reference's (tile | z bits) order with its tie rule (emission order = id
// order): point_list and ranges are bit-identical to the general chain's, which stays as the path of debug mode, of grids of
// more than 32 768 tiles, and of large scenes with very long lists (more than 20 480 instances in a list and 8 Mi in all: the 331k
// trained cloud; the chain notices after its scan, continues on the general chain's un-hinted branch -- the preprocess is not
// repeated -- and the thread remembers the (P, grid) for which that happened).  A list beyond one workgroup's capacity (8192 entries:
// trained clouds, a million Gaussians) is sorted by several workgroups, each a range of the list's (tile, z) axis; a Gaussian of more
// than 256 tiles (a trained scene's background blobs have thousands) is walked by its whole wave, a row per lane.
//
//     1. voxel_cull_count_kernel (voxel_geom.hip)    the part of the preprocess the binning needs (radii, tile cube, z bits) + an LDS
//                                                    histogram of the workgroup's instances over the lists -> H[wg][list]
//     2. voxel_scan_records_kernel (voxel_geom.hip)  two jobs in one launch: exclusive prefix of every column of H, list totals, longest
//                                                    list -- the last scan workgroup posts {num_rendered, longest list} to the host
//                                                    mailbox --, and the rest of the preprocess (the render kernels' records), which
//                                                    runs while the host waits for the totals and sizes the binning / image state
//                                                    (the reference's D2H, VOX/voxelizer_impl.cu:248)
//     4. vox_stick_scatter_kernel                    instances -> list segments; first instance of every Gaussian (the backward's
//                                                    moment rows); workgroup 0 builds the sort kernel's lists
//     5. vox_stick_sort_kernel                       per-list sort, point_list, tiles, ranges
//     6. launch_build_work (binning.hip)             the render kernel's work list from the ranges, as in the general chain
#include "voxel_state.hpp"
#include "dispatch.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <vector>

R2_TS_DEFINE(sticks)
#define VS_TS(ph) R2_TS_AT(sticks, ph)

namespace r2 {

namespace {

constexpr int VS_THREADS = (int)VS_PRODUCER;
constexpr uint32_t VS_MAX_SHIFT = 3;       // tiles per stick = 2^shift <= 8: three bits above the 29-bit id
constexpr uint32_t VS_ID_BITS = 29;
constexpr uint32_t VS_ID_MASK = (1u << VS_ID_BITS) - 1u;

// the sort kernel: 1024-thread workgroups of two kinds, as in raster_tilefirst.hip ("big": one list of up to VSK_BIG_CAP
// entries; "group": four lists of up to VSK_SMALL_CAP entries, one per 256 threads, in lockstep)
constexpr int VSK_THREADS = 1024;
constexpr int VSK_GROUPS = VSK_THREADS / 256;
constexpr uint32_t VSK_SMALL_CAP = 1536, VSK_SMALL_PER = 6, VSK_SMALL_BINS = 1024;
constexpr uint32_t VSK_BIG_CAP = 8192, VSK_BIG_PER = 8, VSK_BIG_BINS = 2048;
constexpr uint32_t VSK_BIG_TARGET = 5120;   // lists beyond VSK_BIG_CAP: ceil(n / 5120) parts, each a range of the list's coarse histogram
constexpr uint32_t VSK_COARSE = 1024;       // bins of that histogram (one per thread)
constexpr size_t VSK_LDS = VSK_BIG_CAP * sizeof(unsigned long long) + (VSK_BIG_BINS + 1 + VSK_COARSE + 1) * sizeof(uint32_t);
static_assert(VSK_COARSE == VSK_THREADS && VSK_BIG_CAP >= VSK_BIG_TARGET + VSK_BIG_TARGET / 2, "parts need slack over their target size");
static_assert(VSK_GROUPS * VSK_SMALL_CAP * sizeof(unsigned long long) + VSK_GROUPS * (VSK_SMALL_BINS + 1) * sizeof(uint32_t) <= VSK_LDS,
              "the groups fit the big layout");
static_assert(VSK_SMALL_PER * 256 == VSK_SMALL_CAP && VSK_BIG_PER * VSK_THREADS == VSK_BIG_CAP, "entries per thread");
static_assert((VSK_SMALL_BINS >> VS_MAX_SHIFT) >= 64, "enough buckets per tile of a stick");

struct __attribute__((aligned(8))) Pair2 { uint2 a, b; };   // two list entries, stored at once wherever the first one lies
struct Cube { uint32_t lx, ly, lz, rw, rh, rd; };
__device__ __forceinline__ Cube cube_of(const uint4 c, uint32_t tt)
{
    Cube q;
    q.lx = c.y & 0xFFFFu; q.ly = c.y >> 16; q.lz = c.z & 0xFFFFu; q.rw = c.z >> 16; q.rh = c.w;
    q.rd = tt / (q.rw * q.rh);   // tiles_touched = rw * rh * rd
    return q;
}

constexpr uint32_t VSS_SERVICE = 2;   // workgroups of the scatter kernel that do not scatter
// ---- 4. scatter.  Workgroup 0 does not scatter: it builds the sort kernel's two lists from the list totals and gives the
// tiles of empty lists their (0, 0) ranges (the reference's memset).
__global__ void __launch_bounds__(VS_THREADS) vox_stick_scatter_kernel(
    int P, uint32_t per_wg, uint32_t ni, uint32_t gx, uint32_t gy, uint32_t T, uint32_t sh, uint32_t NL, uint32_t stride,
    const uint32_t *__restrict__ tiles_touched, uint4 *__restrict__ cube, const uint32_t *__restrict__ depth_key,
    uint32_t *__restrict__ first, uint32_t *__restrict__ order, const uint32_t *__restrict__ H,
    const uint32_t *__restrict__ totals, const uint32_t *__restrict__ wgtot, uint2 *__restrict__ pairs,
    uint2 *__restrict__ ranges, uint4 *__restrict__ big, uint4 *__restrict__ small, uint32_t *__restrict__ nparts,
    uint32_t *__restrict__ work_partial, uint32_t n_partial, const WorkListOut wo /* ranges == nullptr: not here */, uint32_t big_cap)
{
    extern __shared__ uint32_t s_pos[];   // [stride] start of the list's segment + this workgroup's offset in it, bumped per instance
    __shared__ uint32_t s_wsum[2 + VS_PER_THREAD_MAX][VS_THREADS / 64], s_carry[3], s_base;
    static_assert(VS_PER_THREAD_MAX >= 1, "the service workgroup uses three rows");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x == 0) {
        if (tid < 3) s_carry[tid] = 0u;
        for (uint32_t i = tid; i < n_partial; i += VS_THREADS) work_partial[i] = 0u;   // the sort kernel adds to them
        __syncthreads();
        for (uint32_t base = 0; base < NL; base += VS_THREADS) {
            const uint32_t l = base + (uint32_t)tid;
            const uint32_t c = l < NL ? totals[l] : 0u;
            const uint32_t nb = c > VSK_SMALL_CAP ? (c > VSK_BIG_CAP ? (c + VSK_BIG_TARGET - 1u) / VSK_BIG_TARGET : 1u) : 0u;
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
                if (bstart + q < big_cap) big[bstart + q] = make_uint4(l, q, start, c);
            if (ns) small[sstart] = make_uint4(l, 0u, start, c);
            if (l < NL && c == 0u)
                for (uint32_t sub = 0; sub < (1u << sh); ++sub)
                    if ((l << sh) + sub < T) ranges[(l << sh) + sub] = make_uint2(0u, 0u);
            __syncthreads();
            if (tid == VS_THREADS - 1) { s_carry[0] = start + c; s_carry[1] = bstart + nb; s_carry[2] = sstart + ns; }
            __syncthreads();
        }
        if (tid == 0) { nparts[0] = min(s_carry[1], big_cap); nparts[1] = s_carry[2]; }
        return;
    }
    if (blockIdx.x == 1) {
        // ---- second service workgroup.  Grids of up to 4096 tiles: a list IS a tile, so the list totals are all it takes for the
        // tile ranges and the render kernel's work list (the launch of its own that builds them from the ranges otherwise: 9 us)
        if (wo.ranges != nullptr) ranges_and_work_block<VS_THREADS>(totals, wo);
        return;
    }
    const uint32_t wg = blockIdx.x - VSS_SERVICE;
    VS_TS(4);
    // this thread's Gaussians (vs_grid): requested now, used after the scans (everything here was written by earlier kernels on
    // other XCDs)
    constexpr int NI = (int)VS_PER_THREAD_MAX;
    const uint32_t g0 = wg * per_wg, g1 = min(g0 + per_wg, (uint32_t)P);
    uint32_t g_idx[NI], g_tt[NI], g_key[NI];
    uint4 g_cb[NI];
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        g_idx[it] = g0 + (uint32_t)it * (uint32_t)VS_THREADS + (uint32_t)tid;
        const bool own = (uint32_t)it < ni && g_idx[it] < g1;
        const uint32_t gc = min(g_idx[it], (uint32_t)P - 1u);
        g_tt[it] = tiles_touched[gc]; g_cb[it] = cube[gc]; g_key[it] = depth_key[gc];
        g_tt[it] = own ? g_tt[it] : 0u;
        if (own) order[g_idx[it]] = g_idx[it];   // the ids the geometry backward walks: all of them (DW_NVIS = 0)
    }
    // rows of producer workgroups before this one (the first instance of its first Gaussian)
    uint32_t pre = 0;
    for (uint32_t w = tid; w < wg; w += VS_THREADS) pre += wgtot[w];
    // ---- exclusive scan of the list totals (every workgroup for itself: <= 16 KB, cheaper than a launch boundary)
    const uint32_t *__restrict__ my_off = H + (size_t)wg * stride;
    constexpr int MAXQ = (int)(VS_MAX_LISTS / VS_THREADS);
    uint32_t offs[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const uint32_t t = min((uint32_t)(q * VS_THREADS + tid), stride - 1u);
        offs[q] = my_off[t];
        s_pos[t] = totals[t];   // (clamped duplicates store the same value)
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) pre += (uint32_t)__shfl_xor(pre, d);
    if (lane == 0) s_wsum[1][wave] = pre;
    __syncthreads();
    const uint32_t ipt = (stride + VS_THREADS - 1) / VS_THREADS;
    const uint32_t t0 = min(stride, (uint32_t)tid * ipt), t1 = min(t0 + ipt, stride);
    uint32_t sum = 0;
    for (uint32_t t = t0; t < t1; ++t) sum += s_pos[t];
    uint32_t incl = sum, itt[NI];
#pragma unroll
    for (int it = 0; it < NI; ++it) itt[it] = g_tt[it];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const uint32_t ut = __shfl_up(itt[it], d);
            if (lane >= d) itt[it] += ut;
        }
    }
    if (lane == 63) {
        s_wsum[0][wave] = incl;
#pragma unroll
        for (int it = 0; it < NI; ++it) s_wsum[2 + it][wave] = itt[it];
    }
    if (tid == 0) {
        uint32_t b = 0;
#pragma unroll
        for (int w = 0; w < VS_THREADS / 64; ++w) b += s_wsum[1][w];
        s_base = b;
    }
    __syncthreads();
    uint32_t run = incl - sum;
    for (int w = 0; w < wave; ++w) run += s_wsum[0][w];
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t c = s_pos[t];
        s_pos[t] = run;
        run += c;
    }
    // first moment row of each of this thread's Gaussians: the workgroup's Gaussians in id order (it = 0 first)
    uint32_t firstv[NI], before = s_base;
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        uint32_t f = before + itt[it] - g_tt[it], all = 0;
#pragma unroll
        for (int w = 0; w < VS_THREADS / 64; ++w) {
            const uint32_t x = s_wsum[2 + it][w];
            f += w < wave ? x : 0u;
            all += x;
        }
        firstv[it] = f;
        before += all;
    }
    __syncthreads();
    // ... + where this workgroup's instances start inside each segment
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const uint32_t t = (uint32_t)(q * VS_THREADS + tid);
        if (t < stride) s_pos[t] += offs[q];
    }
    __syncthreads();
    VS_TS(5);
    const uint32_t smask = (1u << sh) - 1u;
    // one row of a Gaussian's tile cube (tiles t0r .. t0r + rw - 1: one stick, sometimes more): one LDS atomic per (row, stick), and
    // its instances leave two to a 16-byte store.  The walk is bound by scattered store transactions: ~19 us for a workgroup's 19 k
    // instances whatever the number of workgroups per CU.  Non-temporal stores (the dirty lines then do not wait in the L2s for the
    // end of the kernel): 38 -> 123 us -- the L2's write combining is what makes this affordable
    auto emit_row = [&](uint32_t g, uint32_t key, uint32_t t0r, uint32_t rw) {
        const uint32_t t1r = t0r + rw - 1u;
        for (uint32_t l = t0r >> sh; l <= (t1r >> sh); ++l) {
            const uint32_t a = max(t0r, l << sh), b = min(t1r, (l << sh) + smask);
            const uint32_t cnt = b - a + 1u;
            const uint32_t pos = atomicAdd(&s_pos[l], cnt);
            uint32_t k = 0;
            for (; k + 1u < cnt; k += 2u) {
                Pair2 w;
                w.a = make_uint2(key, g | (((a + k) & smask) << VS_ID_BITS));
                w.b = make_uint2(key, g | (((a + k + 1u) & smask) << VS_ID_BITS));
                *reinterpret_cast<Pair2 *>(pairs + pos + k) = w;
            }
            if (k < cnt) pairs[pos + k] = make_uint2(key, g | (((a + k) & smask) << VS_ID_BITS));
        }
    };
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        // a Gaussian of MANY tiles is not walked by its own lane (a trained scene holds a few of thousands of tiles: one lane
        // walked for 235 us while a workgroup's median was 26) but, below, by the whole wave
        const bool big = g_tt[it] > VS_BIG_GAUSSIAN;
        if (g_tt[it] != 0u) {
            const uint32_t g = g_idx[it];
            // what the backward needs per Gaussian: its run of moment rows (any disjoint assignment serves: here id order)
            first[g] = firstv[it];
            reinterpret_cast<uint32_t *>(cube + g)[0] = firstv[it];
            if (!big) {
                const Cube q = cube_of(g_cb[it], g_tt[it]);
                for (uint32_t z = 0; z < q.rd; ++z)
                    for (uint32_t y = 0; y < q.rh; ++y) emit_row(g, g_key[it], ((q.lz + z) * gy + q.ly + y) * gx + q.lx, q.rw);
            }
        }
        unsigned long long todo = __ballot(big);
        while (todo) {   // (wave-uniform) one such Gaussian at a time, a row per lane
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            const uint32_t og = (uint32_t)__shfl((int)g_idx[it], src), okey = (uint32_t)__shfl((int)g_key[it], src),
                           ott = (uint32_t)__shfl((int)g_tt[it], src);
            const uint4 ocb = make_uint4(0u, (uint32_t)__shfl((int)g_cb[it].y, src), (uint32_t)__shfl((int)g_cb[it].z, src),
                                         (uint32_t)__shfl((int)g_cb[it].w, src));
            const Cube q = cube_of(ocb, ott);
            const uint32_t rows = q.rd * q.rh;
            for (uint32_t r = (uint32_t)lane; r < rows; r += 64u) {
                const uint32_t z = r / q.rh, y = r - z * q.rh;
                emit_row(og, okey, ((q.lz + z) * gy + q.ly + y) * gx + q.lx, q.rw);
            }
        }
    }
    VS_TS(6);
}

// ---- 5. per-list sort
__device__ __forceinline__ unsigned long long vs_pack(const uint2 e)
{
    return ((unsigned long long)(e.y >> VS_ID_BITS) << 61) | ((unsigned long long)e.x << VS_ID_BITS) | (unsigned long long)(e.y & VS_ID_MASK);
}

// key range of a group's entries -> every thread of the group: (kmin, kmax) over all keys, pmax = largest key with a clear sign
// bit, nmin = smallest key with the sign bit set; two workgroup barriers
template <int NT>
__device__ __forceinline__ void vs_group_range(uint32_t &kmin, uint32_t &kmax, uint32_t &pmax, uint32_t &nmin,
                                               uint32_t (*s_mm)[VSK_THREADS / 64], int lane, int wave, int w0)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor(kmin, d));
        kmax = max(kmax, (uint32_t)__shfl_xor(kmax, d));
        pmax = max(pmax, (uint32_t)__shfl_xor(pmax, d));
        nmin = min(nmin, (uint32_t)__shfl_xor(nmin, d));
    }
    __syncthreads();
    if (lane == 0) { s_mm[0][wave] = kmin; s_mm[1][wave] = kmax; s_mm[2][wave] = pmax; s_mm[3][wave] = nmin; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        kmin = min(kmin, s_mm[0][w0 + w]); kmax = max(kmax, s_mm[1][w0 + w]);
        pmax = max(pmax, s_mm[2][w0 + w]); nmin = min(nmin, s_mm[3][w0 + w]);
    }
    // (the same in every lane of a wave: scalar registers -- the sort below runs at the 64-VGPR limit)
    kmin = __builtin_amdgcn_readfirstlane(kmin); kmax = __builtin_amdgcn_readfirstlane(kmax);
    pmax = __builtin_amdgcn_readfirstlane(pmax); nmin = __builtin_amdgcn_readfirstlane(nmin);
}

struct VSWork { uint32_t *partial; uint32_t block_tiles, chunk, min_len; };   // partial == nullptr: the caller counts the work items itself

// mine[u] (entry u * NT + gtid of the list's cnt entries, in registers) -> its sorted position by (tile in stick, z bits, id).
// One-level bucket sort as in raster_tilefirst.hip (tf_sort_group): the BINS buckets are divided among the stick's 2^sh tiles,
// and inside a tile laid over the list's key range.  The keys are raw float bits of world z, compared as unsigned (quirk Q10:
// negative z after positive): a list that straddles z = 0 holds two clusters 2^31 apart, so the buckets of the sign-bit class
// follow those of the other class directly (monotone: the order of the buckets is the order of the keys, whatever the
// distribution; position = bucket base + rank inside the bucket by the whole 64-bit entry, exact).
// The barriers are the WORKGROUP's: every group of the workgroup calls this together.
template <int NT, uint32_t PER, uint32_t BINS>
__device__ __forceinline__ void vs_sort_group(unsigned long long (&mine)[PER], unsigned long long *s_a, uint32_t *s_bin,
                                              uint32_t *s_wsum, int gtid, int lane, int wave, int w0, uint32_t cnt, uint32_t kmin,
                                              uint32_t kmax, uint32_t pmax, uint32_t nmin, uint32_t sh, uint32_t list, uint32_t T,
                                              uint32_t start, uint32_t *__restrict__ point_list, uint32_t *__restrict__ tiles_out,
                                              uint2 *__restrict__ ranges, const VSWork wk,
                                              const uint32_t *s_wb = nullptr /* the entries are one PART of a longer list: [2^sh + 1] first
                                              position of every tile in the whole list */, uint32_t out_off = 0u /* the part's */,
                                              bool first_part = true)
{
    const uint32_t BPS = BINS >> sh;   // buckets per tile of the stick
    // position of a key on the list's axis: its VALUE (the centres of a list's Gaussians are spread evenly in z: linear in the
    // bit pattern crowded the lists around z = 0, whose keys span several binades, into a tenth of their buckets -- the rank loop
    // of such a workgroup ran 17 us against a median of 2), magnitudes capped at FLT_MAX (inf / NaN patterns are the largest of
    // their class: still monotone); the sign-bit class continues where the other one ends
    auto fv = [](uint32_t k) { return __uint_as_float(min(k & 0x7FFFFFFFu, 0x7F7FFFFFu)); };
    const bool both = kmin < 0x80000000u && kmax >= 0x80000000u;
    const float f0 = fv(kmin), fn0 = fv(nmin), noff = both ? fv(pmax) - f0 : 0.f;
    const float tmax = both ? noff + (fv(kmax) - fn0) : fv(kmax) - f0;
    const float scale = tmax > 0.f ? (float)(BPS - 1u) / tmax : 0.f;
    VS_TS(8);
    for (uint32_t i = gtid; i <= BINS; i += NT) s_bin[i] = 0u;
    __syncthreads();
    // (64 VGPRs keep two of these workgroups on a CU: bucket and ticket share a word, and so do a bucket's base and length)
    static_assert(BINS <= (1u << 16) && PER * NT <= (1u << 16), "two 16-bit halves");
    uint32_t bin_ticket[PER];
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
        const uint32_t i = u * NT + (uint32_t)gtid;
        bin_ticket[u] = 0u;
        if (i < cnt) {
            const uint32_t key = (uint32_t)(mine[u] >> VS_ID_BITS), sub = (uint32_t)(mine[u] >> 61);
            const float t = (both && key >= 0x80000000u) ? noff + (fv(key) - fn0) : fv(key) - f0;
            const uint32_t bin = sub * BPS + min((uint32_t)(t * scale), BPS - 1u);
            bin_ticket[u] = bin | (atomicAdd(&s_bin[bin], 1u) << 16);
        }
    }
    __syncthreads();
    VS_TS(9);
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
    VS_TS(10);
    // the ranges of the stick's tiles: their buckets' first and one-past-last positions (identifyTileRanges; empty tiles keep (0, 0))
    // ... and, for the render kernel's work list (launch_build_work_from_partials), the stick's work items added to the count of
    // its block of tiles (a stick never straddles two blocks; the scatter kernel zeroed the counts)
    if ((s_wb != nullptr ? first_part : cnt != 0u) && (uint32_t)gtid < (1u << sh)) {
        const uint32_t tile = (list << sh) + (uint32_t)gtid;
        uint32_t items = 0u;
        if (tile < T) {
            const uint32_t a = s_wb ? s_wb[gtid] : s_bin[(uint32_t)gtid * BPS], b = s_wb ? s_wb[gtid + 1] : s_bin[((uint32_t)gtid + 1u) * BPS];
            ranges[tile] = b > a ? make_uint2(start + a, start + b) : make_uint2(0u, 0u);
            const uint32_t ch = work_tile_chunk(wk.chunk, b - a);
            items = (b - a) < wk.min_len ? 0u : (b - a + ch - 1u) / ch;
        }
        if (wk.partial != nullptr) {
            for (uint32_t d = 1; d < (1u << sh); d <<= 1) items += (uint32_t)__shfl_xor(items, (int)d);   // lanes 0 .. 2^sh - 1: all here
            if (gtid == 0 && items != 0u) atomicAdd(&wk.partial[(list << sh) / wk.block_tiles], items);
        }
    }
    uint32_t base_len[PER];
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
        const uint32_t i = u * NT + (uint32_t)gtid;
        base_len[u] = 0u;
        if (i < cnt) {
            const uint32_t bin = bin_ticket[u] & 0xFFFFu;
            const uint32_t b0 = s_bin[bin], b1 = s_bin[bin + 1u];
            base_len[u] = b0 | ((b1 - b0) << 16);
            if (b1 - b0 > 1u) s_a[b0 + (bin_ticket[u] >> 16)] = mine[u];   // a bucket of one needs neither the store nor a rank
        }
    }
    __syncthreads();
    VS_TS(11);
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
        const uint32_t i = u * NT + (uint32_t)gtid;
        if (i < cnt) {
            const uint32_t b0 = base_len[u] & 0xFFFFu, len = base_len[u] >> 16;
            uint32_t r = 0;
            if (len > 1u)
                for (uint32_t q = b0; q < b0 + len; ++q) r += s_a[q] < mine[u] ? 1u : 0u;
            base_len[u] = b0 + r;   // the entry's position in the sorted list
        }
    }
    // The ids go to their positions in LDS first and leave in list order with unit-stride stores: written straight from the
    // lanes that hold them, every store instruction of a wave touched 64 different cache lines -- two scattered 4-byte stores per
    // instance were most of this kernel (66 -> see DESIGN.md section 4).  The tile of position j follows from the tile boundaries.
    uint32_t *s_out = reinterpret_cast<uint32_t *>(s_a);
    __syncthreads();   // the ranks have been read
    VS_TS(12);
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
        const uint32_t i = u * NT + (uint32_t)gtid;
        if (i < cnt) s_out[base_len[u]] = (uint32_t)mine[u] & VS_ID_MASK;
    }
    uint32_t bnd[(1u << VS_MAX_SHIFT) - 1u];   // first position of tiles 1 .. 2^sh - 1 of the stick
#pragma unroll
    for (uint32_t k = 0; k < (1u << VS_MAX_SHIFT) - 1u; ++k)
        bnd[k] = (k + 1u) < (1u << sh) ? __builtin_amdgcn_readfirstlane(s_wb ? s_wb[k + 1u] : s_bin[(k + 1u) * BPS]) : 0xFFFFFFFFu;
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
        const uint32_t j = u * NT + (uint32_t)gtid;
        if (j < cnt) {
            const uint32_t pos = out_off + j;   // position in the whole list
            uint32_t sub = 0;
#pragma unroll
            for (uint32_t k = 0; k < (1u << VS_MAX_SHIFT) - 1u; ++k) sub += pos >= bnd[k] ? 1u : 0u;
            point_list[start + pos] = s_out[j];
            tiles_out[start + pos] = (list << sh) + sub;
        }
    }
    VS_TS(13);
}

// (Measured and left out, round 5: PERSISTENT workgroups, two per CU, each taking items blockIdx.x, + gridDim.x, ... and requesting
//  the next item's descriptor while it sorts the current one -- a workgroup's start is two dependent round trips, 4.2 of its
//  10.7 us.  Inlined, the loop made the compiler hoist every item's address arithmetic in front of it (20 registers spilled); as
//  calls, with the static assignment: 51 -> 82 us.  The hardware's dispatch of one workgroup per item balances lists of 1600 to
//  6000 entries better than a stride does.)
// PARTS: the call holds lists beyond one workgroup's capacity (the host knows: the scan kernel told it the longest list).  A
// variant of its own: with that path compiled in, the kernel spills 28 registers at the 64 its two workgroups per CU allow.
template <bool PARTS>
__global__ void __launch_bounds__(VSK_THREADS, 8) vox_stick_sort_kernel(
    const uint4 *__restrict__ big, const uint4 *__restrict__ small, const uint32_t *__restrict__ nparts,
    const uint2 *__restrict__ pairs, uint32_t sh, uint32_t T, uint32_t *__restrict__ point_list, uint32_t *__restrict__ tiles_out,
    uint2 *__restrict__ ranges, const VSWork wk, uint32_t big_cap)
{
    extern __shared__ unsigned long long vsk_lds[];
    __shared__ uint32_t s_mm[4][VSK_THREADS / 64], s_wsum[VSK_THREADS / 64], s_wb[(1u << VS_MAX_SHIFT) + 1u], s_cnt, s_off;
    const uint32_t p = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the descriptor is requested together with the counts that say whether it exists (one round trip, not two)
    VS_TS(7);
    const uint4 pd = big[min(p, big_cap - 1u)];
    const uint32_t nbig = nparts[0], nsmall = nparts[1];
    if (p >= nbig) {
        // ---- group: VSK_GROUPS short lists, one per 256 threads of the workgroup
        const uint32_t q0 = (p - nbig) * (uint32_t)VSK_GROUPS;
        if (q0 >= nsmall) return;
        const int g = tid >> 8, gtid = tid & 255, w0 = g * 4;
        unsigned long long *s_a = vsk_lds + (size_t)g * VSK_SMALL_CAP;
        uint32_t *s_bin = reinterpret_cast<uint32_t *>(vsk_lds + VSK_GROUPS * VSK_SMALL_CAP) + (size_t)g * (VSK_SMALL_BINS + 1);
        uint32_t n = 0, start = 0, list = 0;
        if (q0 + (uint32_t)g < nsmall) {
            const uint4 sd = small[q0 + g];
            list = sd.x; start = sd.z; n = sd.w;
        }
        list = __builtin_amdgcn_readfirstlane(list); start = __builtin_amdgcn_readfirstlane(start); n = __builtin_amdgcn_readfirstlane(n);
        uint32_t kmin = 0xFFFFFFFFu, kmax = 0u, pmax = 0u, nmin = 0xFFFFFFFFu;
        unsigned long long mine[VSK_SMALL_PER];
        {
            // all loads in flight before the first use, branch-free (clamped addresses)
            uint2 e[VSK_SMALL_PER];
            const uint32_t last = start + (n ? n - 1u : 0u);
#pragma unroll
            for (uint32_t u = 0; u < VSK_SMALL_PER; ++u) e[u] = pairs[min(start + u * 256u + (uint32_t)gtid, last)];
#pragma unroll
            for (uint32_t u = 0; u < VSK_SMALL_PER; ++u) {
                const bool in = u * 256u + (uint32_t)gtid < n;
                const bool neg = (e[u].x >> 31) != 0u;
                mine[u] = in ? vs_pack(e[u]) : ~0ull;
                kmin = in ? min(kmin, e[u].x) : kmin;
                kmax = in ? max(kmax, e[u].x) : kmax;
                pmax = (in && !neg) ? max(pmax, e[u].x) : pmax;
                nmin = (in && neg) ? min(nmin, e[u].x) : nmin;
            }
        }
        vs_group_range<256>(kmin, kmax, pmax, nmin, s_mm, lane, wave, w0);
        vs_sort_group<256, VSK_SMALL_PER, VSK_SMALL_BINS>(mine, s_a, s_bin, s_wsum, gtid, lane, wave, w0, n, kmin, kmax, pmax, nmin, sh,
                                                          list, T, start, point_list, tiles_out, ranges, wk);
        return;
    }
    // ---- big: one list (or one part of a very long one) for the whole workgroup
    unsigned long long *s_a = vsk_lds;                                             // [VSK_BIG_CAP]
    uint32_t *s_bin = reinterpret_cast<uint32_t *>(s_a + VSK_BIG_CAP);             // [VSK_BIG_BINS + 1]
    uint32_t *s_coarse = s_bin + VSK_BIG_BINS + 1;                                 // [VSK_COARSE + 1]
    const uint32_t list = __builtin_amdgcn_readfirstlane(pd.x), part = __builtin_amdgcn_readfirstlane(pd.y),
                   start = __builtin_amdgcn_readfirstlane(pd.z), n = __builtin_amdgcn_readfirstlane(pd.w);
    const uint2 *__restrict__ src = pairs + start;
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u, pmax = 0u, nmin = 0xFFFFFFFFu;
    unsigned long long mine[VSK_BIG_PER];
    if (!PARTS || n <= VSK_BIG_CAP) {
        uint2 e[VSK_BIG_PER];
#pragma unroll
        for (uint32_t u = 0; u < VSK_BIG_PER; ++u) e[u] = src[min(u * VSK_THREADS + (uint32_t)tid, n - 1u)];
#pragma unroll
        for (uint32_t u = 0; u < VSK_BIG_PER; ++u) {
            const bool in = u * VSK_THREADS + (uint32_t)tid < n;   // (u * VSK_THREADS + tid < VSK_BIG_CAP anyway)
            const bool neg = (e[u].x >> 31) != 0u;
            mine[u] = in ? vs_pack(e[u]) : ~0ull;
            kmin = in ? min(kmin, e[u].x) : kmin;
            kmax = in ? max(kmax, e[u].x) : kmax;
            pmax = (in && !neg) ? max(pmax, e[u].x) : pmax;
            nmin = (in && neg) ? min(nmin, e[u].x) : nmin;
        }
        vs_group_range<VSK_THREADS>(kmin, kmax, pmax, nmin, s_mm, lane, wave, 0);
        vs_sort_group<VSK_THREADS, VSK_BIG_PER, VSK_BIG_BINS>(mine, s_a, s_bin, s_wsum, tid, lane, wave, 0, min(n, VSK_BIG_CAP), kmin, kmax,
                                                              pmax, nmin, sh, list, T, start, point_list, tiles_out, ranges, wk);
        return;
    }
    if (!PARTS) return;
    // ---- this part's share of a list no workgroup can hold: a contiguous range of the bins of a COARSE histogram of the list over
    // its (tile in stick, z) axis, chosen so that the parts are balanced whatever the distribution (raster_tilefirst.hip does the
    // same over depth).  The bins are laid out from every 8th entry -- every part of the list takes the same sample, hence the
    // same layout; the layout only balances the parts: the bin of an entry is a monotone function of (tile, key), so the parts
    // are consecutive pieces of the sorted list, and membership and offsets below are exact.
    constexpr uint32_t SAMPLE = 8;
    const uint32_t nsub = 1u << sh, CB = VSK_COARSE >> sh;   // coarse bins per tile
    const uint32_t nparts_list = (n + VSK_BIG_TARGET - 1u) / VSK_BIG_TARGET, ns = (n + SAMPLE - 1u) / SAMPLE;
    for (uint32_t base = 0; base < ns; base += 4u * VSK_THREADS) {   // key range of the sample: four loads in flight, branch-free
        uint32_t kk[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) kk[u] = src[min((base + u * VSK_THREADS + (uint32_t)tid) * SAMPLE, n - 1u)].x;
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u)
            if (base + u * VSK_THREADS + (uint32_t)tid < ns) {
                const bool neg = (kk[u] >> 31) != 0u;
                kmin = min(kmin, kk[u]); kmax = max(kmax, kk[u]);
                pmax = neg ? pmax : max(pmax, kk[u]);
                nmin = neg ? min(nmin, kk[u]) : nmin;
            }
    }
    vs_group_range<VSK_THREADS>(kmin, kmax, pmax, nmin, s_mm, lane, wave, 0);
    // coarse bin of an entry: tile * CB + position of its key's VALUE in the sample's range (vs_sort_group has the reasoning);
    // keys outside that range, and keys of a sign class the sample has not seen, clamp into the end bins of where they belong
    auto fv = [](uint32_t k) { return __uint_as_float(min(k & 0x7FFFFFFFu, 0x7F7FFFFFu)); };
    const bool has_pos = kmin < 0x80000000u, has_neg = kmax >= 0x80000000u;
    const float f0 = fv(has_pos ? kmin : 0u), fn0 = fv(has_neg ? nmin : 0u);
    const float noff = (has_pos && has_neg) ? fv(pmax) - f0 : 0.f;                        // where the sign-bit class starts
    const float ptop = has_pos ? (has_neg ? noff : fv(kmax) - f0) : 0.f;                  // where the other class ends
    const float tmax = has_neg ? noff + (fv(kmax) - fn0) : ptop;
    const float cscale = tmax > 0.f ? (float)CB / tmax : 0.f;
    auto coarse_of = [&](const uint2 e) {
        const uint32_t key = e.x, sub = e.y >> VS_ID_BITS;
        float t;
        if (key >= 0x80000000u) t = has_neg ? noff + fmaxf(fv(key) - fn0, 0.f) : tmax;   // after every key with a clear sign bit
        else t = has_pos ? fminf(fmaxf(fv(key) - f0, 0.f), ptop) : 0.f;                   // before every key with the sign bit set
        return sub * CB + min((uint32_t)(t * cscale), CB - 1u);
    };
    for (uint32_t i = tid; i <= VSK_COARSE; i += VSK_THREADS) s_coarse[i] = 0u;
    if (tid <= (int)(1u << VS_MAX_SHIFT)) s_wb[tid] = 0u;
    if (tid == 0) { s_cnt = 0u; s_off = 0u; }
    __syncthreads();
    for (uint32_t base = 0; base < ns; base += 4u * VSK_THREADS) {
        uint2 ee[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) ee[u] = src[min((base + u * VSK_THREADS + (uint32_t)tid) * SAMPLE, n - 1u)];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u)
            if (base + u * VSK_THREADS + (uint32_t)tid < ns) atomicAdd(&s_coarse[coarse_of(ee[u])], 1u);
    }
    __syncthreads();
    {
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
        const uint32_t owner = min((uint32_t)(((unsigned long long)excl * nparts_list) / ns), nparts_list - 1u);
        __syncthreads();
        s_coarse[tid] = owner;   // from here on: the part that owns the bin
        __syncthreads();
    }
    // ---- one pass over the whole list: entries of parts before mine are counted (my offset in the sorted list), mine appended,
    // and every tile's entries counted (the tile ranges; which tile an output position belongs to)
    kmin = 0xFFFFFFFFu; kmax = 0u; pmax = 0u; nmin = 0xFFFFFFFFu;   // reused for this part's own key range
    uint32_t below = 0, wsub[1u << VS_MAX_SHIFT];
#pragma unroll
    for (uint32_t k = 0; k < (1u << VS_MAX_SHIFT); ++k) wsub[k] = 0u;
    for (uint32_t base4 = 0; base4 < n; base4 += 4u * VSK_THREADS) {   // whole waves stay in the loop: the append is wave-cooperative
        uint2 e4[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) e4[u] = src[min(base4 + u * VSK_THREADS + (uint32_t)tid, n - 1u)];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t i = base4 + u * VSK_THREADS + (uint32_t)tid;
            const uint2 e = e4[u];
            const bool in = i < n;
            const uint32_t own = in ? s_coarse[coarse_of(e)] : 0xFFFFFFFFu;
            below += own < part ? 1u : 0u;
            const uint32_t sub = e.y >> VS_ID_BITS;
#pragma unroll
            for (uint32_t k = 0; k < (1u << VS_MAX_SHIFT); ++k)
                if (k < nsub) wsub[k] += (uint32_t)__popcll(__ballot(in && sub == k));   // (per wave: the same in all its lanes)
            const bool take = own == part;
            const unsigned long long mm = __ballot(take);
            if (mm) {
                uint32_t wbase = 0;
                const int leader = __ffsll((long long)mm) - 1;
                if (lane == leader) wbase = atomicAdd(&s_cnt, (uint32_t)__popcll(mm));
                wbase = __shfl(wbase, leader);
                const uint32_t slot = wbase + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
                if (take && slot < VSK_BIG_CAP) {
                    s_a[slot] = vs_pack(e);
                    const bool neg = (e.x >> 31) != 0u;
                    kmin = min(kmin, e.x); kmax = max(kmax, e.x);
                    pmax = neg ? pmax : max(pmax, e.x);
                    nmin = neg ? min(nmin, e.x) : nmin;
                }
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) below += __shfl_xor(below, d);
    if (lane == 0) {
        atomicAdd(&s_off, below);
#pragma unroll
        for (uint32_t k = 0; k < (1u << VS_MAX_SHIFT); ++k)
            if (k < nsub) atomicAdd(&s_wb[k + 1u], wsub[k]);
    }
    __syncthreads();
    if (tid == 0) {   // counts -> first position of every tile in the list
        uint32_t run = 0;
        for (uint32_t k = 1; k <= nsub; ++k) { run += s_wb[k]; s_wb[k] = run; }
    }
    __syncthreads();
    const uint32_t cnt = s_cnt, out_off = s_off;
    if (cnt > VSK_BIG_CAP) {
        // (more entries inside one sliver of the axis than the LDS holds: equal z.)  Rank by counting, straight from memory:
        // O(cnt x n).  Exact like everything else.
        if (part == 0u && (uint32_t)tid < nsub) {
            const uint32_t tile = (list << sh) + (uint32_t)tid;
            if (tile < T) {
                const uint32_t a = s_wb[tid], b = s_wb[tid + 1];
                ranges[tile] = b > a ? make_uint2(start + a, start + b) : make_uint2(0u, 0u);
                const uint32_t ch = work_tile_chunk(wk.chunk, b - a);
                const uint32_t items = (b - a) < wk.min_len ? 0u : (b - a + ch - 1u) / ch;
                if (wk.partial != nullptr && items != 0u) atomicAdd(&wk.partial[(list << sh) / wk.block_tiles], items);
            }
        }
        for (uint32_t i = tid; i < n; i += VSK_THREADS) {
            const uint2 e = src[i];
            if (s_coarse[coarse_of(e)] != part) continue;
            const unsigned long long me = vs_pack(e);
            uint32_t r = 0;
            for (uint32_t j = 0; j < n; ++j) {
                const uint2 o = src[j];
                if (s_coarse[coarse_of(o)] == part && vs_pack(o) < me) ++r;
            }
            point_list[start + out_off + r] = e.y & VS_ID_MASK;
            tiles_out[start + out_off + r] = (list << sh) + (e.y >> VS_ID_BITS);
        }
        return;
    }
    vs_group_range<VSK_THREADS>(kmin, kmax, pmax, nmin, s_mm, lane, wave, 0);   // (its barriers also order the appends before the reads)
#pragma unroll
    for (uint32_t u = 0; u < VSK_BIG_PER; ++u) {
        const uint32_t i = u * VSK_THREADS + (uint32_t)tid;
        mine[u] = i < cnt ? s_a[i] : ~0ull;
    }
    __syncthreads();   // everybody holds its entries: s_a may be overwritten by the placement
    vs_sort_group<VSK_THREADS, VSK_BIG_PER, VSK_BIG_BINS>(mine, s_a, s_bin, s_wsum, tid, lane, wave, 0, cnt, kmin, kmax, pmax, nmin, sh,
                                                          list, T, start, point_list, tiles_out, ranges, wk, s_wb, out_off, part == 0u);
}

// ---- host side
// what the thread has learnt about a (P, grid): a stick list too long for the sort kernel -> the general chain from then on
// The note ages: after VS_NOTE_RETRY declined calls the chain is tried again -- the cloud behind a (P, grid) changes
// while it trains -- and r2_voxel_sticks_limits drops every thread's notes (a process-wide epoch).
struct VSNote { int P, nx, ny, nz; bool bad; unsigned long long used; uint32_t declined; };
thread_local std::vector<VSNote> g_vs_notes;
thread_local unsigned long long g_vs_tick = 0;
thread_local int g_vs_notes_epoch = 0;
std::atomic<int> g_vs_epoch{0};
constexpr uint32_t VS_NOTE_RETRY = 64;

VSNote *vs_note(int P, const VoxelGrid &v, bool create)
{
    if (const int e = g_vs_epoch.load(std::memory_order_relaxed); e != g_vs_notes_epoch) {   // the limits changed: start over
        g_vs_notes.clear();
        g_vs_notes_epoch = e;
    }
    for (VSNote &n : g_vs_notes)
        if (n.P == P && n.nx == v.nx && n.ny == v.ny && n.nz == v.nz) { n.used = ++g_vs_tick; return &n; }
    if (!create) return nullptr;
    if (g_vs_notes.size() >= 32) {
        size_t lru = 0;
        for (size_t i = 1; i < g_vs_notes.size(); ++i)
            if (g_vs_notes[i].used < g_vs_notes[lru].used) lru = i;
        g_vs_notes.erase(g_vs_notes.begin() + (long)lru);
    }
    g_vs_notes.push_back(VSNote{P, v.nx, v.ny, v.nz, false, ++g_vs_tick, 0u});
    return &g_vs_notes.back();
}

// forwards that took the chain, forwards that left it after the scan (a list too long), forwards it declined
std::atomic<long long> g_vs_taken{0}, g_vs_fallback{0}, g_vs_declined{0};
// a call whose longest list exceeds the first AND whose instances exceed the second goes to the general chain (r2_voxel_sticks_limits)
constexpr long long VS_LONG_LIST = 4 * 5120, VS_LONG_SCENE = 8ll << 20;
std::atomic<long long> g_vs_long_list{VS_LONG_LIST}, g_vs_long_scene{VS_LONG_SCENE};
std::atomic<int> g_vs_no_parts{0};   // tests: treat a list beyond one workgroup's capacity as unsupported (the fallback's path)
std::atomic<int> g_vs_mode{-1};   // -1: not decided yet (environment), 0: off, 1: every grid it can serve

int vs_mode()
{
    int m = g_vs_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        const char *e = getenv("R2_VOXEL_STICKS");
        m = (e && e[0] == '0') ? 0 : 1;
        g_vs_mode.store(m, std::memory_order_relaxed);
    }
    return m;
}

}  // namespace

bool voxel_sticks_switched_on() { return vs_mode() != 0; }

// The sort kernel needs VSK_LDS = 77.8 KB of dynamic LDS per workgroup: a device whose opt-in limit is below that (64 KB parts)
// cannot run the chain -- voxel_forward_choice then reports voxel.general.device_lds (r2_path_stats) and the general chain serves.
bool voxel_sticks_lds_ok()
{
    static signed char lds_state[R2_MAX_DEVICES] = {}, lds_state2[R2_MAX_DEVICES] = {};
    return VSK_LDS <= device_lds_optin_bytes() &&
           allow_dynamic_lds(reinterpret_cast<const void *>(vox_stick_sort_kernel<false>), (int)VSK_LDS, lds_state) &&
           allow_dynamic_lds(reinterpret_cast<const void *>(vox_stick_sort_kernel<true>), (int)VSK_LDS, lds_state2);
}

int voxel_forward_sticks(r2_alloc_fn binningBuffer, void *binning_user, r2_alloc_fn imageBuffer, void *image_user,
                         const VoxelGeom &geom, const VoxelGrid &v, int P, const float *means3D, const float *opacities,
                         const float *scales, float scale_modifier, const float *rotations, const float *cov3D_precomp,
                         float *out_volume, int *radii_x, int *radii_y, int *radii_z, uint32_t sh /* tiles per stick = 2^sh */,
                         hipStream_t s)
{
    const size_t T = (size_t)v.gx * v.gy * v.gz;
    const size_t V = (size_t)v.nx * v.ny * v.nz;
    // (whether a call comes here at all, and its stick width: voxel_forward_choice, dispatch.hpp)
    static_assert(VS_MAX_SHIFT == DISPATCH_VOX_STICK_MAX_SHIFT && VS_MAX_LISTS == DISPATCH_VOX_STICK_LISTS &&
                  ((size_t)1 << VS_ID_BITS) == DISPATCH_VOX_STICK_P, "dispatch.hpp states this chain's limits");
    if (VSNote *n = vs_note(P, v, false); n && n->bad) {
        if (++n->declined < VS_NOTE_RETRY) {
            path_count(PS_VOX_GENERAL_REMEMBERED);
            g_vs_declined.fetch_add(1, std::memory_order_relaxed);
            return VOX_STICKS_NOT_TAKEN;
        }
        n->bad = false;   // try again: the cloud behind this (P, grid) may have changed
        n->declined = 0u;
    }
    const uint32_t NL = (uint32_t)((T + ((size_t)1 << sh) - 1) >> sh);
    uint32_t stride = 32;
    while (stride < NL) stride <<= 1;
    const VoxelSticks st = VoxelSticks::carve(geom.stick_temp, P);
    const VSGrid grid = vs_grid(P, device_cu_count());
    const uint32_t NW = grid.wgs;   // <= st.NW rows
    int dev = 0;
    R2_HIP_TRY(hipGetDevice(&dev));
    VSCounters *ctr = reinterpret_cast<VSCounters *>(voxel_small_counter_block(dev, s));   // (zero-filled first if a call left it dirty)
    if (!ctr) {
        path_count(PS_VOX_GENERAL_NO_WORKSPACE);
        g_vs_declined.fetch_add(1, std::memory_order_relaxed);
        return VOX_STICKS_NOT_TAKEN;
    }
    uint32_t *nparts = st.ctr;

    uint32_t *mailbox = nullptr, seq = 0;
    int rc = host_mailbox_arm(&mailbox, &seq);
    if (rc) return rc;
    { StageScope t(ST_VOX_PREPROCESS, s);
    launch_voxel_cull_count(geom, v, P, grid, means3D, scales, scale_modifier, rotations, cov3D_precomp, radii_x, radii_y, radii_z, sh, stride,
                            st.H, st.wgtot, ctr, s); }
    R2_HIP_TRY(hipGetLastError());
    // one launch for two independent jobs: the column scan (its last workgroup posts the totals to the host) and the render
    // records, which nothing before the render kernel reads -- they are written while the host waits for the totals, sizes the
    // two remaining state buffers and launches the rest (as a part of the first kernel they left the GPU idle for 12 us there; as
    // a launch of their own behind the scan they cost a kernel boundary).  Measured and left out: a first message with the
    // instance count from the cull + count kernel's last workgroup, the rest of the chain enqueued before the scan's verdict on
    // the longest list is read -- the records already cover the wait, and the extra counter cost the first kernel 4 us
    // (475 -> 482 us per query)
    { StageScope t(ST_VOX_SCAN, s);
    launch_voxel_scan_records(geom, v, P, grid, means3D, opacities, cov3D_precomp, st.H, NW, stride, st.totals, ctr, mailbox, seq, s); }
    R2_HIP_TRY(hipGetLastError());
    uint32_t hw[DW_COUNT] = { 0 };
    rc = host_mailbox_wait(seq, hw, DW_COUNT, s);
    if (rc) return rc;
    voxel_counter_block_clean(dev, s);   // the scan's last workgroup has posted the totals and put the zeros back
    const uint32_t num_rendered = hw[DW_TOTAL], longest = hw[DW_PMAX];
    if (num_rendered > 0x7FFFFFFFu) {
        set_error("r2_voxel_forward: %u (tile, Gaussian) instances do not fit the 31-bit num_rendered", num_rendered);
        return R2_ERR_INVALID;
    }
    // lists beyond one workgroup's capacity are sorted in parts: at most NL + R / VSK_BIG_TARGET descriptors of long lists.  More
    // than the state holds (Gaussians of hundreds of tiles each): the general chain, from here and from now on
    const size_t parts_bound = (size_t)NL + (size_t)num_rendered / VSK_BIG_TARGET + 1;
    // ... and so do large scenes with VERY long lists (the 331k trained cloud: 22 M instances, lists of 28 000): every part of such
    // a list reads the whole list, and the part-wise sort then costs more than the general chain's two radix passes -- sort 572 us
    // against 341 + 66 for the depth order, the ranges and the work list; the query 2449 us against 2223.  Long lists in smaller
    // scenes are fine (a million small Gaussians, 8 M instances, lists of 10 000: 1340 -> 1276 us), and so are large Gaussians
    // (the 92k trained cloud, 65 tiles per Gaussian, 6 M instances, longest list 7245: 791 -> 755 us) -- since a Gaussian of
    // thousands of tiles is walked by its whole wave in the count and scatter kernels; before that, one lane walked a background
    // blob's tiles for 235 us and the same query took 964.
    const uint32_t nvis = hw[DW_NVIS];
    static const bool dbg = [] { const char *e = getenv("R2_VOXEL_STICKS_DEBUG"); return e && e[0] == '1'; }();
    if (dbg) fprintf(stderr, "voxel sticks: P %d R %u visible %u longest list %u\n", P, num_rendered, nvis, longest);
    const bool large = (long long)longest > g_vs_long_list.load(std::memory_order_relaxed) &&
                       (long long)num_rendered > g_vs_long_scene.load(std::memory_order_relaxed);
    if (large || parts_bound > st.bigcap || (g_vs_no_parts.load(std::memory_order_relaxed) && longest > VSK_BIG_CAP)) {
        VSNote *n = vs_note(P, v, true);
        n->bad = true;
        n->declined = 0u;
        path_count(PS_VOX_GENERAL_LONG_LISTS);
        g_vs_fallback.fetch_add(1, std::memory_order_relaxed);
        return VOX_STICKS_FALLBACK;
    }
    const size_t R = num_rendered;
    char *bchunk = binningBuffer(VoxelBinning::carve(nullptr, R).bytes, binning_user);
    char *ichunk = imageBuffer(VoxelImage::carve(nullptr, T, V, R, false, vox_chunk_for(v.gy, v.gz)).bytes, image_user);
    if (!bchunk || !ichunk) {
        set_error("r2_voxel_forward: binning/image allocation callback returned NULL");
        return R2_ERR_ALLOC;
    }
    const VoxelBinning bin = VoxelBinning::carve(bchunk, R);
    const VoxelImage img = VoxelImage::carve(ichunk, T, V, R, false, vox_chunk_for(v.gy, v.gz));
    uint2 *pairs = reinterpret_cast<uint2 *>(bin.part);   // the backward's moment scratch (48 bytes per instance), free until then
    // the render kernel's work list: for more than 4096 tiles the sort kernel leaves the work items per block of tiles behind
    // (one launch of the construction instead of two)
    const bool sums = T > 4096 && img.work_temp != nullptr && R > 0;
    const VSWork wk{sums ? reinterpret_cast<uint32_t *>(img.work_temp) : nullptr, build_work_block_tiles(), vox_work_chunk(v.gy, v.gz),
                    voxel_short_list_min(false)};
    const uint32_t n_partial = sums ? (uint32_t)((T + wk.block_tiles - 1) / wk.block_tiles) : 0u;
    // ... and for up to 4096 tiles (lists = tiles) the scatter kernel's second service workgroup builds ranges and work list
    const bool direct = sh == 0u;
    const WorkListOut wo{direct ? img.ranges : nullptr, img.chunk_base, img.work_tile, (uint32_t)T, wk.chunk, nullptr, wk.min_len, 0u};
    { StageScope t(ST_VOX_DUPLICATE, s);
    vox_stick_scatter_kernel<<<dim3(NW + VSS_SERVICE), dim3(VS_THREADS), stride * sizeof(uint32_t), s>>>(
        P, grid.per_wg, grid.ni, (uint32_t)v.gx, (uint32_t)v.gy, (uint32_t)T, sh, NL, stride, geom.tiles_touched, geom.cube, geom.depth_key, geom.first,
        geom.order, st.H, st.totals, st.wgtot, pairs, img.ranges, st.big, st.small, nparts, wk.partial, n_partial, wo, (uint32_t)st.bigcap); }
    R2_HIP_TRY(hipGetLastError());
    if (R > 0) {
        StageScope t(ST_VOX_SORT, s);
        const dim3 sgrid((unsigned)(std::min(parts_bound, st.bigcap) + (NL + VSK_GROUPS - 1) / VSK_GROUPS));
        if (longest > VSK_BIG_CAP)
            vox_stick_sort_kernel<true><<<sgrid, dim3(VSK_THREADS), VSK_LDS, s>>>(st.big, st.small, nparts, pairs, sh, (uint32_t)T, bin.point_list,
                                                                                  bin.tiles, img.ranges, wk, (uint32_t)st.bigcap);
        else
            vox_stick_sort_kernel<false><<<sgrid, dim3(VSK_THREADS), VSK_LDS, s>>>(st.big, st.small, nparts, pairs, sh, (uint32_t)T, bin.point_list,
                                                                                   bin.tiles, img.ranges, wk, (uint32_t)st.bigcap);
    }
    R2_HIP_TRY(hipGetLastError());
    if (!direct) {
    StageScope t(ST_VOX_RANGES, s);
    if (sums) launch_build_work_from_partials(img.ranges, (uint32_t)T, wk.chunk, img.chunk_base, img.work_tile, wk.partial, s, wk.min_len);
    else launch_build_work(img.ranges, (uint32_t)T, wk.chunk, img.chunk_base, img.work_tile, img.work_temp, s, wk.min_len); }
    R2_HIP_TRY(hipGetLastError());
    { StageScope t(ST_VOX_RENDER_FWD, s);
    launch_voxel_render_forward(geom, bin, img, v, out_volume, false, s); }
    R2_HIP_TRY(hipGetLastError());
    g_vs_taken.fetch_add(1, std::memory_order_relaxed);
    path_count(PS_VOX_STICK_FIRST);
    return (int)num_rendered;
}

void voxel_sticks_release() { g_vs_notes.clear(); }

}  // namespace r2

extern "C" void r2_voxel_sticks_stats(long long *out, int reset)
{
    std::atomic<long long> *c[3] = {&r2::g_vs_taken, &r2::g_vs_fallback, &r2::g_vs_declined};
    for (int i = 0; i < 3; ++i) {
        if (out) out[i] = c[i]->load(std::memory_order_relaxed);
        if (reset) c[i]->store(0, std::memory_order_relaxed);
    }
}

extern "C" void r2_voxel_sticks_limits(long long longest_list, long long instances)
{
    r2::g_vs_long_list.store(longest_list > 0 ? longest_list : r2::VS_LONG_LIST, std::memory_order_relaxed);
    r2::g_vs_long_scene.store(instances > 0 ? instances : r2::VS_LONG_SCENE, std::memory_order_relaxed);
    r2::g_vs_epoch.fetch_add(1, std::memory_order_relaxed);   // what the threads remember was decided under the old limits
}

extern "C" void r2_voxel_sticks_control(int mode)
{
    if (mode == 0 || mode == 1) r2::g_vs_mode.store(mode, std::memory_order_relaxed);
    if (mode == 3) r2::g_vs_notes.clear();   // the calling thread's notes
    if (mode == 4 || mode == 5) r2::g_vs_no_parts.store(mode == 4 ? 1 : 0, std::memory_order_relaxed);   // tests
}
