// voxel_small.hip -- the voxelizer's binning for SMALL grids (<= 64 tiles: the training loop's 32^3 TV patch, train.py:128-142).
//
// Reference: VOX/voxelizer_impl.cu:171-302 (duplicateWithKeys -> SortPairs -> identifyTileRanges).  The general pipeline
// (voxel_api.hip) reproduces that chain for any grid with a depth order over ALL Gaussians, an emission pass and a tile sort --
// nine launches whose cost is latency, not work, when 2 % of the Gaussians reach the patch.  Here:
//   1. voxel_preprocess_small_kernel (voxel_geom.hip): the preprocess + a compact list of the survivors
//      {id, depth key, tile cube, first scratch row}, totals to the host mailbox;
//   2. voxel_small_lists_kernel (below): ONE workgroup per tile walks the survivor list, keeps the entries whose cube holds
//      its tile, sorts them by (depth key, id) in LDS -- the reference's stable (tile | depth) order, bit for bit -- and writes
//      its segment of point_list; every workgroup also counts ALL tiles (64 LDS counters), so it knows its segment's offset and
//      the tile ranges without a scan kernel; workgroup 0 builds the render kernels' work list;
//   3. the render kernels of the general path (voxel_render.hip), unchanged.
// The host round trip (num_rendered sizes the binning / image state, VOX/voxelizer_impl.cu:248) sits between 1 and 3 and hides
// behind 2: the lists are built in a temp inside the geometry state before the other two buffers exist, the render kernels read
// them there, and a small copy kernel moves them into the binning / image state afterwards (for the backward, which is the
// general one: it needs ranges, point_list, the per-Gaussian cube record and the scratch rows, all of which exist).  6 launches
// instead of 11, none of them over more than the survivors except the preprocess.  Totals beyond what the path holds (LDS sort,
// temp sizes) send the call back to the general path.
#include "voxel_state.hpp"
#include "dispatch.hpp"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace r2 {

namespace {

constexpr int SL_THREADS = 1024;
constexpr int SB_BINS = 4096;                                              // buckets of the in-LDS sort
constexpr int SB_PER_THREAD = (int)(VOX_SMALL_MAX_SURVIVORS / SL_THREADS);   // entries per thread at the largest list
constexpr size_t SL_LDS_BYTES = 2 * (size_t)VOX_SMALL_MAX_SURVIVORS * sizeof(unsigned long long) + (SB_BINS + 1) * sizeof(uint32_t);

// Where the lists are built before the host has sized the binning / image state: the radix-fallback temp of the depth order
// inside the geometry state (>= 12 bytes per Gaussian, unused on this path).
struct SmallTmp {
    uint2 *ranges;          // [64]
    uint32_t *chunk_base;   // [64 + 2]
    uint32_t *counts;       // [64] entries per tile
    uint4 *work;            // [cap_work]
    uint32_t *plist;        // [cap_R] sorted Gaussian ids, tile after tile
    uint32_t *tiles;        // [cap_R] tile of every sorted instance (the backward's per-instance tile id)
    uint32_t cap_work, cap_R;
    static SmallTmp carve(char *base, size_t bytes)
    {
        SmallTmp t;
        // header: ranges [0, 512), chunk_base [512, 776), counts [1024, 1280) -- disjoint (chunk_base[T] is written too)
        constexpr size_t OFF_CHUNK = 512, OFF_COUNTS = 1024, HEADER = 1280;
        static_assert(VOX_SMALL_MAX_TILES * sizeof(uint2) <= OFF_CHUNK, "ranges overlap chunk_base");
        static_assert(OFF_CHUNK + (VOX_SMALL_MAX_TILES + 2) * sizeof(uint32_t) <= OFF_COUNTS, "chunk_base overlaps counts");
        static_assert(OFF_COUNTS + VOX_SMALL_MAX_TILES * sizeof(uint32_t) <= HEADER, "counts overlap the work list");
        t.ranges = reinterpret_cast<uint2 *>(base);
        t.chunk_base = reinterpret_cast<uint32_t *>(base + OFF_CHUNK);
        t.counts = reinterpret_cast<uint32_t *>(base + OFF_COUNTS);
        const size_t avail = bytes > HEADER ? bytes - HEADER : 0;
        t.cap_work = (uint32_t)std::min<size_t>(8192, avail / 128);   // 1/8 of the space: an item stands for >= 128 instances
        t.work = reinterpret_cast<uint4 *>(base + HEADER);
        t.plist = reinterpret_cast<uint32_t *>(base + HEADER + (size_t)t.cap_work * 16);
        t.cap_R = (uint32_t)std::min<size_t>((avail - (size_t)t.cap_work * 16) / 8, 0x7FFFFFFFu);
        t.tiles = t.plist + t.cap_R;
        return t;
    }
};

__global__ void __launch_bounds__(SL_THREADS) voxel_small_lists_kernel(
    const uint4 *__restrict__ surv, const uint32_t *__restrict__ words /* DW_TOTAL, DW_NVIS: written by the preprocess */,
    int gx, int gy, int gz, uint32_t T, SmallTmp tmp, uint32_t min_len, uint32_t *arrivals /* zero between calls */)
{
    extern __shared__ unsigned long long s_keys[];   // [VOX_SMALL_MAX_SURVIVORS] (depth key << 32) | id of this tile's entries
    __shared__ uint32_t s_cnt, s_low[SL_THREADS / 64], s_last;
    __shared__ uint32_t s_counts[VOX_SMALL_MAX_TILES];
    const uint32_t t = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nsurv = words[DW_NVIS], R = words[DW_TOTAL];
    // more survivors / instances than this path holds: the host sees the same totals and runs the general path instead
    if (nsurv > VOX_SMALL_MAX_SURVIVORS || R > tmp.cap_R) return;
    if (tid == 0) s_cnt = 0u;
    __syncthreads();
    // Every survivor's tile cube as a 64-bit mask over the (<= 64) tiles: membership of THIS tile is bit t, and the number of
    // instances in the tiles before it -- this tile's offset in point_list -- is the sum of popcount(mask & bits below t): no
    // per-tile histogram, no scan kernel.
    const unsigned long long below = t ? ((t >= 64u) ? ~0ull : ((1ull << t) - 1ull)) : 0ull;
    uint32_t low = 0;
    constexpr int PF = 4;   // batches whose loads are in flight together (the loop was one dependent load round trip per batch)
    for (uint32_t base0 = 0; base0 < nsurv; base0 += PF * SL_THREADS) {
    uint4 pre[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const uint32_t i = base0 + (uint32_t)(u * SL_THREADS + tid);
        pre[u] = surv[min(i, nsurv - 1u)];
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {   // whole waves stay in the loop: the append below is wave-cooperative
        const uint32_t i = base0 + (uint32_t)(u * SL_THREADS + tid);
        if (base0 + (uint32_t)(u * SL_THREADS) >= nsurv) break;   // workgroup-uniform
        const uint4 e = i < nsurv ? pre[u] : make_uint4(0u, 0u, 0u, 0u);   // an empty cube: member of nothing
        const int lx = (int)(e.z & 15u), ly = (int)((e.z >> 4) & 15u), lz = (int)((e.z >> 8) & 15u);
        const int hx = (int)((e.z >> 12) & 15u), hy = (int)((e.z >> 16) & 15u), hz = (int)((e.z >> 20) & 15u);
        const unsigned long long xbits = ((1ull << (hx - lx)) - 1ull) << lx;
        unsigned long long mask = 0ull;
        for (int z = lz; z < hz; ++z)
            for (int y = ly; y < hy; ++y) mask |= xbits << ((z * gy + y) * gx);
        low += (uint32_t)__popcll(mask & below);
        // append this tile's members: one LDS atomic per wave (ballot + prefix popcount), not one per member
        const bool member = (mask >> t) & 1ull;
        const unsigned long long mm = __ballot(member);
        if (mm) {
            uint32_t wbase = 0;
            const int leader = __ffsll((long long)mm) - 1;
            if (lane == leader) wbase = atomicAdd(&s_cnt, (uint32_t)__popcll(mm));
            wbase = __shfl(wbase, leader);
            if (member) s_keys[wbase + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull))] = ((unsigned long long)e.y << 32) | (unsigned long long)e.x;
        }
    }
    }
    (void)gz;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) low += __shfl_xor(low, d);
    if (lane == 0) s_low[wave] = low;
    __syncthreads();
    const uint32_t cnt = s_cnt;
    uint32_t off = 0;
#pragma unroll
    for (int w = 0; w < SL_THREADS / 64; ++w) off += s_low[w];
    // ---- this tile's count, for whoever builds the ranges + work list
    // (agent-scope store: it goes past the non-coherent per-XCD L2s, so the workgroup that builds the work list sees it without a
    // release fence -- at agent scope a fence writes back the XCD's whole L2, see raster_render.hip)
    if (tid == 0) __hip_atomic_store(&tmp.counts[t], cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- sort by (depth key, id).  One-level bucket sort in LDS (the depth keys are float bit patterns of world z inside one
    // tile's reach: a narrow range, spread evenly): ~2 entries per value-linear bucket, an entry's position = its bucket's base +
    // its rank among the bucket's entries by (key, id) -- exact whatever the distribution (many equal keys just make a long
    // bucket).  Eight workgroup barriers; a 1024-thread bitonic network needs 20-28 (measured: 16 us of this kernel's 31).
    unsigned long long *s_sorted = s_keys + VOX_SMALL_MAX_SURVIVORS;   // second half of the dynamic LDS
    uint32_t *s_bin = reinterpret_cast<uint32_t *>(s_sorted + VOX_SMALL_MAX_SURVIVORS);   // [SB_BINS + 1] counts -> bases
    __shared__ uint32_t s_mm[2][SL_THREADS / 64], s_wsum[SL_THREADS / 64];
    {
        // key range of this tile's entries (as unsigned bit patterns: the sort order; negative z sorts after positive, Q10)
        uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
        for (uint32_t i = tid; i < cnt; i += SL_THREADS) {
            const uint32_t kk = (uint32_t)(s_keys[i] >> 32);
            kmin = min(kmin, kk);
            kmax = max(kmax, kk);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            kmin = min(kmin, (uint32_t)__shfl_xor(kmin, d));
            kmax = max(kmax, (uint32_t)__shfl_xor(kmax, d));
        }
        if (lane == 0) { s_mm[0][wave] = kmin; s_mm[1][wave] = kmax; }
        for (uint32_t i = tid; i <= SB_BINS; i += SL_THREADS) s_bin[i] = 0u;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < SL_THREADS / 64; ++w) { kmin = min(kmin, s_mm[0][w]); kmax = max(kmax, s_mm[1][w]); }
        // bucket = monotone in the unsigned key: linear in the bit pattern (within one sign class the pattern is monotone in the
        // value; a patch that straddles z = 0 just gets uneven buckets)
        const float scale = kmax > kmin ? (float)(SB_BINS - 1) / (float)(kmax - kmin) : 0.f;
        auto bucket_of = [&](uint32_t kk) { return min((uint32_t)((float)(kk - kmin) * scale), (uint32_t)(SB_BINS - 1)); };
        uint32_t my_bin[SB_PER_THREAD], my_ticket[SB_PER_THREAD];
#pragma unroll
        for (int u = 0; u < SB_PER_THREAD; ++u) {
            const uint32_t i = (uint32_t)(u * SL_THREADS + tid);
            my_bin[u] = 0u; my_ticket[u] = 0u;
            if (i < cnt) {
                my_bin[u] = bucket_of((uint32_t)(s_keys[i] >> 32));
                my_ticket[u] = atomicAdd(&s_bin[my_bin[u]], 1u);
            }
        }
        __syncthreads();
        // exclusive prefix of the bucket counts: SB_BINS / 1024 consecutive buckets per thread
        constexpr int BPT = SB_BINS / SL_THREADS;
        uint32_t c[BPT], tsum = 0;
#pragma unroll
        for (int q = 0; q < BPT; ++q) { c[q] = s_bin[tid * BPT + q]; tsum += c[q]; }
        uint32_t incl = tsum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        uint32_t run = incl - tsum;
        for (int w = 0; w < wave; ++w) run += s_wsum[w];
#pragma unroll
        for (int q = 0; q < BPT; ++q) { s_bin[tid * BPT + q] = run; run += c[q]; }
        if (tid == SL_THREADS - 1) s_bin[SB_BINS] = run;
        __syncthreads();
        // place: bucket base + ticket (arbitrary order inside the bucket), then rank inside the bucket by (key, id)
#pragma unroll
        for (int u = 0; u < SB_PER_THREAD; ++u) {
            const uint32_t i = (uint32_t)(u * SL_THREADS + tid);
            if (i < cnt) s_sorted[s_bin[my_bin[u]] + my_ticket[u]] = s_keys[i];
        }
        __syncthreads();
        unsigned long long mine[SB_PER_THREAD];
        uint32_t dest[SB_PER_THREAD];
#pragma unroll
        for (int u = 0; u < SB_PER_THREAD; ++u) {
            const uint32_t i = (uint32_t)(u * SL_THREADS + tid);
            mine[u] = 0ull; dest[u] = 0xFFFFFFFFu;
            if (i < cnt) {
                mine[u] = s_keys[i];
                const uint32_t b0 = s_bin[my_bin[u]], b1 = s_bin[my_bin[u] + 1u];
                uint32_t r = 0;
                for (uint32_t q = b0; q < b1; ++q) r += s_sorted[q] < mine[u] ? 1u : 0u;
                dest[u] = b0 + r;
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < SB_PER_THREAD; ++u)
            if (dest[u] != 0xFFFFFFFFu) s_keys[dest[u]] = mine[u];
    }
    __syncthreads();
    for (uint32_t i = tid; i < cnt; i += SL_THREADS) {
        tmp.plist[off + i] = (uint32_t)s_keys[i];
        tmp.tiles[off + i] = t;
    }
    // ---- the LAST workgroup to get here builds the render kernels' work list from the 64 tile counts
    __syncthreads();
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the count has been acknowledged: it is visible device-wide
        s_last = (atomicAdd(arrivals, 1u) == T - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    if (tid == 0) *arrivals = 0u;   // ready for the next call
    if (tid < (int)VOX_SMALL_MAX_TILES)
        s_counts[tid] = (uint32_t)tid < T ? __hip_atomic_load(&tmp.counts[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    __syncthreads();
    // tile ranges (empty tiles stay (0, 0) like the reference's zero-filled array, VOX/voxelizer_impl.cu:289) + work list
    // no short-list exemption: the item kernel's own short-list branch serves the few short tiles of a patch, which spares the
    // short-list kernel's launch.  (ONE item per tile -- no partial sums at all -- was measured: the 128 long-running workgroups
    // took 38 us instead of 23.)
    (void)min_len;
    ranges_and_work_block<SL_THREADS>(s_counts, WorkListOut{tmp.ranges, tmp.chunk_base, tmp.work, T, vox_work_chunk(gy, gz), nullptr, 0u,
                                                            tmp.cap_work});
}

// one 64-byte device word per host thread: the survivor / row counter of the small-grid preprocess, zero between calls
// The two persistent counters of the path ({done | survivors | rows} of the preprocess, `arrivals` of the lists kernel) are
// self-resetting, so they must not be shared by two calls in flight: one 64-byte block per (host thread, device, stream), kept
// for the life of the thread (a thread that alternates devices or streams finds its block again instead of allocating).
// one 64-byte counter block per (host thread, device, stream), owned by the thread: freed when it exits or on
// r2_thread_release(); a thread that cycles through more streams than the table holds evicts the least recently used entry
// (round 4: never freed, and the path switched itself off for the thread after 256 streams -- ADVICE r4)
// dirty: a chain armed the block's counters and has not yet seen them reset by its own kernels (set before the first kernel that
// bumps them, cleared after the host has read the totals): found dirty by the next call -- one that aborted half way -- the block is
// zero-filled first (the stick chain's counters had no such recovery)
struct SmallCounter { int dev; hipStream_t stream; unsigned long long *ptr; unsigned long long used; bool dirty; };
struct SmallCounters {
    std::vector<SmallCounter> v;
    unsigned long long tick = 0;
    static void free_one(const SmallCounter &c)
    {
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return; }
        if (cur != c.dev && hipSetDevice(c.dev) != hipSuccess) { (void)hipGetLastError(); return; }
        if (hipFree(c.ptr) != hipSuccess) (void)hipGetLastError();   // (waits for the device: its last call may still run)
        if (cur != c.dev) (void)hipSetDevice(cur);
    }
    void release()
    {
        for (const SmallCounter &c : v) free_one(c);
        v.clear();
    }
    ~SmallCounters() { release(); }
};
thread_local SmallCounters g_small_counters;
constexpr size_t SMALL_MAX_COUNTERS = 16;

unsigned long long *small_counter_for(int dev, hipStream_t s)
{
    SmallCounters &t = g_small_counters;
    ++t.tick;
    for (SmallCounter &c : t.v)
        if (c.dev == dev && c.stream == s) {
            c.used = t.tick;
            if (c.dirty) {
                if (hipMemsetAsync(c.ptr, 0, 64, s) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            }
            c.dirty = true;   // until voxel_counter_block_clean()
            return c.ptr;
        }
    if (t.v.size() >= SMALL_MAX_COUNTERS) {
        size_t lru = 0;
        for (size_t i = 1; i < t.v.size(); ++i)
            if (t.v[i].used < t.v[lru].used) lru = i;
        SmallCounters::free_one(t.v[lru]);
        t.v.erase(t.v.begin() + (long)lru);
    }
    unsigned long long *p = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&p), 64) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (hipMemsetAsync(p, 0, 64, s) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(p);
        return nullptr;
    }
    t.v.push_back(SmallCounter{dev, s, p, t.tick, true});
    return p;
}

}  // namespace

// the calling thread's block for (dev, s) has been through a whole call: its kernels have put the zeros back
void voxel_counter_block_clean(int dev, hipStream_t s)
{
    for (SmallCounter &c : g_small_counters.v)
        if (c.dev == dev && c.stream == s) c.dirty = false;
}

bool voxel_small_switched_on()
{
    static const bool on = [] { const char *e = getenv("R2_VOXEL_SMALL"); return !(e && e[0] == '0'); }();   // R2_VOXEL_SMALL=0: off
    return on;
}
bool voxel_small_lds_ok()
{
    // more than 64 KB of LDS per workgroup (gfx950: 160 KB per CU) has to be asked for
    static signed char lds_state[R2_MAX_DEVICES] = {};
    return SL_LDS_BYTES <= device_lds_optin_bytes() &&
           allow_dynamic_lds(reinterpret_cast<const void *>(voxel_small_lists_kernel), (int)SL_LDS_BYTES, lds_state);
}

int voxel_forward_small(r2_alloc_fn binningBuffer, void *binning_user, r2_alloc_fn imageBuffer, void *image_user,
                        const VoxelGeom &geom, const VoxelGrid &v, int P, const float *means3D, const float *opacities,
                        const float *scales, float scale_modifier, const float *rotations, const float *cov3D_precomp,
                        float *out_volume, int *radii_x, int *radii_y, int *radii_z, hipStream_t s)
{
    const size_t T = (size_t)v.gx * v.gy * v.gz;
    const size_t V = (size_t)v.nx * v.ny * v.nz;
    // the preprocess packs {workgroups done : 12 | survivors : 20 | rows : 32} into one 64-bit atomic: P < 2^20 keeps every field
    // inside its bits whatever the scene (survivors <= P, workgroups = P / 1024 < 2^10, rows <= 64 tiles x P < 2^26)
    // (x-slab calls: the survivor kernel below rebuilds the tile cube from the radii, without the slab's clip -- general chain)
    // (whether a call comes here at all: voxel_forward_choice, dispatch.hpp)
    int dev = 0;
    R2_HIP_TRY(hipGetDevice(&dev));
    unsigned long long *const g_small_counter = small_counter_for(dev, s);
    if (!g_small_counter) {
        path_count(PS_VOX_GENERAL_NO_WORKSPACE);
        return VOX_SMALL_NOT_TAKEN;
    }
    uint4 *surv = depth_order_slots(geom.dorder_temp, (size_t)P);
    const SmallTmp tmp = SmallTmp::carve(geom.psort_temp, geom.psort_bytes);
    uint32_t *mailbox = nullptr, seq = 0;
    int rc = host_mailbox_arm(&mailbox, &seq);
    if (rc) return rc;
    { StageScope t(ST_VOX_PREPROCESS, s);
    launch_voxel_preprocess_small(geom, v, P, means3D, scales, scale_modifier, rotations, opacities, cov3D_precomp, radii_x, radii_y,
                                  radii_z, surv, g_small_counter, mailbox, seq, s); }
    // the lists are built while the host is still waiting for the totals and sizing the two remaining state buffers
    { StageScope t(ST_VOX_SORT, s);
    voxel_small_lists_kernel<<<dim3((unsigned)T), dim3(SL_THREADS), SL_LDS_BYTES, s>>>(
        surv, geom.host_words, v.gx, v.gy, v.gz, (uint32_t)T, tmp, voxel_short_list_min(false),
        reinterpret_cast<uint32_t *>(g_small_counter + 1)); }
    R2_HIP_TRY(hipGetLastError());
    uint32_t hw[DW_COUNT] = { 0 };
    rc = host_mailbox_wait(seq, hw, DW_COUNT, s);
    if (rc) return rc;
    const uint32_t num_rendered = hw[DW_TOTAL], nsurv = hw[DW_NVIS];
    const size_t R = num_rendered;
    if (num_rendered > 0x7FFFFFFFu) {
        set_error("r2_voxel_forward: %u (tile, Gaussian) instances do not fit the 31-bit num_rendered", num_rendered);
        return R2_ERR_INVALID;
    }
    const size_t NW = R / vox_chunk_for(v.gy, v.gz) + T;
    // rare (a patch that most Gaussians reach): the general path; the kernel above has seen the same totals and done nothing
    voxel_counter_block_clean(dev, s);   // (the preprocess's last workgroup has reset the counter; the lists kernel resets its own)
    if (nsurv > VOX_SMALL_MAX_SURVIVORS || R > tmp.cap_R || NW > tmp.cap_work) {
        path_count(PS_VOX_GENERAL_SMALL_OVERFLOW);
        return VOX_SMALL_NOT_TAKEN;
    }
    path_count(PS_VOX_SMALL_GRID);
    char *bchunk = binningBuffer(VoxelBinning::carve(nullptr, R).bytes, binning_user);
    char *ichunk = imageBuffer(VoxelImage::carve(nullptr, T, V, R, false, vox_chunk_for(v.gy, v.gz)).bytes, image_user);
    if (!bchunk || !ichunk) {
        set_error("r2_voxel_forward: binning/image allocation callback returned NULL");
        return R2_ERR_ALLOC;
    }
    const VoxelBinning bin = VoxelBinning::carve(bchunk, R);
    const VoxelImage img = VoxelImage::carve(ichunk, T, V, R, false, vox_chunk_for(v.gy, v.gz));
    { StageScope t(ST_VOX_RENDER_FWD, s);
    VoxelBinning b2 = bin;          // the render kernels read the lists where they were built ...
    VoxelImage i2 = img;
    b2.point_list = tmp.plist;
    i2.ranges = tmp.ranges;
    i2.chunk_base = tmp.chunk_base;
    i2.work_tile = tmp.work;
    // ... and the last of them also moves them into the binning / image state, where the backward and the introspection look
    VoxelPublish pub{tmp.plist, tmp.tiles, tmp.chunk_base, tmp.ranges, tmp.work, bin.point_list, bin.tiles, img.chunk_base, img.ranges,
                     img.work_tile, (uint32_t)R, (uint32_t)T, (uint32_t)NW, 0u};
    pub.n = std::max(std::max(pub.R, pub.T + 1u), pub.NW);
    launch_voxel_render_forward(geom, b2, i2, v, out_volume, false, s, /*no_short_kernel=*/true, &pub); }
    R2_HIP_TRY(hipGetLastError());
    return (int)num_rendered;
}

void voxel_small_release() { g_small_counters.release(); }

unsigned long long *voxel_small_counter_block(int dev, hipStream_t s) { return small_counter_for(dev, s); }

}  // namespace r2
