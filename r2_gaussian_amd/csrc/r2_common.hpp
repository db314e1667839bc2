// r2_common.hpp -- shared host/device helpers for libr2hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include "../../include/r2hip.h"

namespace r2 {

constexpr int TILE2D = 16;       // 16x16 pixel tiles  (reference BLOCK_X/BLOCK_Y, RAS/config.h:16-17)
constexpr int TILE3D = 8;        // 8x8x8 voxel tiles  (reference BLOCK3D_*, VOX/config.h:16-18)
constexpr int WAVE = 64;         // CDNA4 wavefront
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

void set_error(const char *fmt, ...);

#define R2_HIP_TRY(expr)                                                                         \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            r2::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return -(int)_e;                                                                      \
        }                                                                                         \
    } while (0)

// the reference's CHECK_CUDA(A, debug): sync + check after a stage, only in debug mode
#define R2_STAGE_CHECK(debug, stream, what)                                                      \
    do {                                                                                          \
        hipError_t _e = hipGetLastError();                                                        \
        if (_e == hipSuccess && (debug)) _e = hipStreamSynchronize(stream);                       \
        if (_e != hipSuccess) {                                                                   \
            r2::set_error("stage '%s' failed: %s", what, hipGetErrorString(_e));                  \
            return -(int)_e;                                                                      \
        }                                                                                         \
    } while (0)

// ---- experiment builds only (-DR2_EXP_TS): s_memrealtime (100 MHz) stamps per phase and workgroup, one table per
// translation unit, read back by scripts/cbench through r2_debug_ts_<unit>() -- the timeline inside and between kernels
#ifdef R2_EXP_TS
#define R2_TS_DEFINE(unit)                                                                                              \
    namespace r2 { __device__ unsigned long long g_ts_##unit[16][2048]; }                                               \
    extern "C" __attribute__((visibility("default"))) int r2_debug_ts_##unit(unsigned long long *out)                   \
    {                                                                                                                   \
        return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(r2::g_ts_##unit), sizeof(unsigned long long) * 16 * 2048);      \
    }
// slots 0..2046: the stamp of that workgroup (larger grids: only their first 2047 workgroups); slot 2047: the latest stamp
// of every 32nd workgroup (atomic max) -- the end of a phase over the whole grid
#define R2_TS_AT(unit, ph)                                                                                              \
    do {                                                                                                                \
        if (threadIdx.x == 0) {                                                                                         \
            const unsigned long long _t = wall_clock64();                                                               \
            if (blockIdx.x < 2047u) r2::g_ts_##unit[ph][blockIdx.x] = _t;                                               \
            if ((blockIdx.x & 31u) == 31u) atomicMax(&r2::g_ts_##unit[ph][2047], _t);                                                                  \
        }                                                                                                               \
    } while (0)
// the same for kernels of many short workgroups: every 8th workgroup stamps (slot = workgroup / 8), so that 16 k of them are covered
#define R2_TS_AT8(unit, ph)                                                                                             \
    do {                                                                                                                \
        if (threadIdx.x == 0 && (blockIdx.x & 7u) == 0u && (blockIdx.x >> 3) < 2047u)                                   \
            r2::g_ts_##unit[ph][blockIdx.x >> 3] = wall_clock64();                                                      \
    } while (0)
#else
#define R2_TS_DEFINE(unit)
#define R2_TS_AT(unit, ph)
#define R2_TS_AT8(unit, ph)
#endif

// ---- optional per-stage timing with HIP events on the caller's stream (r2_profile_* in r2hip.h)
enum Stage {
    ST_RAS_PREPROCESS = 0, ST_RAS_SCAN, ST_RAS_DUPLICATE, ST_RAS_SORT, ST_RAS_RANGES, ST_RAS_RENDER_FWD,
    ST_RAS_RENDER_BWD, ST_RAS_GEOM_BWD,
    ST_VOX_PREPROCESS, ST_VOX_SCAN, ST_VOX_DUPLICATE, ST_VOX_SORT, ST_VOX_RANGES, ST_VOX_RENDER_FWD,
    ST_VOX_RENDER_BWD, ST_VOX_GEOM_BWD, ST_KNN, ST_RAS_DEPTHSORT, ST_VOX_DEPTHSORT, ST_COUNT
};
extern int g_profile_mask_on;
void stage_begin(int stage, hipStream_t s);
void stage_end(int stage, hipStream_t s);
struct StageScope {
    int st; hipStream_t s; bool on;
    StageScope(int stage, hipStream_t stream) : st(stage), s(stream), on(g_profile_mask_on != 0) { if (on) stage_begin(st, s); }
    ~StageScope() { if (on) stage_end(st, s); }
};

// 128-byte aligned bump allocation inside a caller-provided chunk (reference: obtain(), RAS/rasterizer_impl.h:21-31)
struct Bump {
    char *base;
    size_t off;
    explicit Bump(char *b) : base(b), off(0) {}
    template <typename T>
    T *take(size_t count)
    {
        off = (off + 127) & ~size_t(127);
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
    size_t offset_of_next() const { return (off + 127) & ~size_t(127); }
    size_t total() const { return off + 128; }
};

// Work items of a tile list: `chunk` instances each -- or, with WORK_CHUNK_ADAPT set on the chunk argument (round 6, the voxelizer),
// a chunk the LIST chooses: at least the base chunk, and long lists are cut into about eight pieces (a power of two, at most 1024).
// What a list is cut into then depends on the grid's base chunk and on its own length only: the partial sums of a tile associate the
// same way in an x-slab call and in the unsharded one (bit-identical volumes, voxel_state.hpp), while a dense small grid (the 32^3
// TV patch of a trained cloud: 3 100 instances per tile) no longer pays for 25 items per tile because a sparse one wants 128.
constexpr uint32_t WORK_CHUNK_ADAPT = 0x80000000u;
__host__ __device__ inline uint32_t work_tile_chunk(uint32_t chunk, uint32_t len)
{
    if (!(chunk & WORK_CHUNK_ADAPT)) return chunk;
    const uint32_t base = chunk & ~WORK_CHUNK_ADAPT;
#ifndef R2_WORK_CHUNK_SHIFT
#define R2_WORK_CHUNK_SHIFT 3   /* pieces of a long list: 2^3 (measured against 4 and 16: profiles/r06e_voxel_adaptive_chunk_ab.txt) */
#endif
    const uint32_t eighth = (len + (1u << R2_WORK_CHUNK_SHIFT) - 1u) >> R2_WORK_CHUNK_SHIFT;
    if (eighth <= base) return base;   // (every list of a 256^3 query: no further arithmetic)
    const uint32_t want = eighth >= 1024u ? 1024u : 1u << (32 - __builtin_clz(eighth - 1u));   // the next power of two
    return want > base ? want : base;
}

// ---- binning.hip: scan / stable radix sort / tile ranges (shared by rasterizer and voxelizer)
size_t scan_temp_bytes(int P);
int inclusive_scan_u32(void *temp, size_t temp_bytes, const uint32_t *in, uint32_t *out, int P, hipStream_t s);
// radix_sort.hip: stable LSD radix sort of (u32 key, u32 value) pairs on key bits [0, end_bit).
// The binning pipeline sorts TWICE instead of once on 64-bit (tile|depth) keys like the reference
// (RAS/rasterizer_impl.cu:301-306): the P Gaussians by depth bits (ties keep ascending id), then the R
// instances -- emitted in that depth order -- by tile id only.  A stable sort by tile of a depth-ordered
// sequence IS the (tile|depth) order with the reference's tie rule, so point_list is bit-identical.
size_t sort_temp_bytes(size_t n);
// extended form: optional second payload (win -> wout), vin == nullptr means "value = input index", allow_skip lets the
// device skip digit passes that are constant over all keys, and for a single-pass sort *totals_out receives a device
// pointer to the per-digit key counts (the bucket sizes), valid until the temp storage is reused.
int sort_pairs_ex(void *temp, size_t temp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
                  const uint32_t *win, uint32_t *wout, size_t n, int end_bit, bool allow_skip, const uint32_t **totals_out,
                  hipStream_t s, uint32_t *inv_out = nullptr /* with win and no skipping: inv_out[value] = position */);
// inclusive scan of in[order[j]] over j (the per-Gaussian tile counts visited in depth order)
// depth_order.hip: ids in (key, id) order by a one-level bucket sort.  depth_order_prepare zeroes the control block + counters
// and is called BEFORE the kernel that produces the keys (which may set the DW_USER word); depth_order_buckets is the
// un-hinted path: it raises DW_OVERFLOW when a bucket was too full (result invalid: fall back to sort_pairs_ex).
size_t depth_order_temp_bytes(size_t P);
int depth_order_prepare(void *temp, size_t temp_bytes, size_t P, hipStream_t s);
int depth_order_buckets(void *temp, size_t temp_bytes, const uint32_t *keys, uint32_t *order, size_t P, hipStream_t s);
// ---- depth order, hinted fast path.  When the caller knows roughly where the keys lie (the range seen by its previous
// call), the kernel that PRODUCES the keys can drop them into value-linear buckets right away: one returning 64-bit atomic per
// key hands out a ticket inside the bucket and adds the Gaussian's instance count to the bucket's total.  One
// dual prefix sum over the buckets then yields both the depth order (bucket base + rank inside the bucket) AND the
// instance offsets in that order, i.e. it replaces min/max + count + scan + place + rank + the offsets scan of the
// un-hinted path (4 launches instead of 9).  The hint only shapes the buckets (keys outside it are clamped into the end
// buckets, which keeps the map monotone): the result is the exact (key, id) order whatever the hint; an over-full bucket
// raises the overflow word and the caller falls back to the un-hinted path.
constexpr uint32_t DEPTH_CULLED_KEY = 0xFFFFFFFFu;
struct DepthHint {
    float plo, pscale;      // keys with a clear sign bit (positive floats): bucket = (value - plo) * pscale, clamped to [0, npos)
    float nlo, nscale;      // keys with the sign bit set (bits grow with |value|): npos + (|value| - nlo) * nscale, clamped
    uint32_t npos, nneg;    // buckets of each class; npos + nneg = number of buckets
};
struct DepthReg {
    unsigned long long *ct; // [nb] zeroed by depth_order_prepare; per bucket: (number of keys << 32) | sum of their instance
                            //      counts, bumped by ONE 64-bit atomic per key.  nullptr = fast path off
    uint2 *bt;              // [P]  {bucket, ticket} of every visible key
    uint32_t *wgmm;         // [4 * producer workgroups] key extrema {pmax, ~pmin, nmax, ~nmin} per workgroup
    DepthHint h;
    uint32_t *payload;      // [P] optional: one word per visible key that travels with it into the sorted records (rasterizer: its
                            //     tile rectangle, depth_rect_pack), which then also gives its instance count.  nullptr = off
};
// tile rectangle of a Gaussian in one word: x0 | y0 << 8 | (w - 1) << 16 | (h - 1) << 24 (grids up to 256 x 256 tiles, w, h >= 1)
__host__ __device__ __forceinline__ uint32_t depth_rect_pack(int x0, int y0, int w, int h)
{
    return (uint32_t)x0 | (uint32_t)y0 << 8 | (uint32_t)(w - 1) << 16 | (uint32_t)(h - 1) << 24;
}
__host__ __device__ __forceinline__ uint32_t depth_rect_count(uint32_t r) { return (((r >> 16) & 0xFFu) + 1u) * ((r >> 24) + 1u); }
__device__ __forceinline__ uint32_t hinted_bucket(uint32_t key, const DepthHint &h)
{
    const bool neg = (key >> 31) != 0u;
    const uint32_t cnt = neg ? h.nneg : h.npos;
    if (cnt == 0u) return neg ? h.npos - 1u : 0u;   // class unseen by the hint: share the neighbouring end bucket
    const float v = __uint_as_float(key & 0x7FFFFFFFu);
    const float x = (v - (neg ? h.nlo : h.plo)) * (neg ? h.nscale : h.pscale);
    // NaN bit patterns are the largest keys of their class: last bucket
    const uint32_t b = (v != v) ? cnt - 1u : (uint32_t)fminf(fmaxf(x, 0.f), (float)(cnt - 1u));
    return (neg ? h.npos : 0u) + b;
}
// Producer side, in two steps so that the atomic's round trip hides behind the rest of the producer's work:
//   depth_register_key  as soon as a visible Gaussian's key and instance count (> 0) are known (any control flow);
//   depth_register_end  by EVERY thread of the 256-thread workgroup at the end (key = DEPTH_CULLED_KEY for threads without
//                       a visible Gaussian): stores {bucket, ticket} and publishes the workgroup's key extrema.
__device__ __forceinline__ uint2 depth_register_key(const DepthReg &r, uint32_t key, uint32_t n_inst)
{
    if (r.ct == nullptr) return make_uint2(0u, 0u);
    const uint32_t b = hinted_bucket(key, r.h);
    const unsigned long long old = atomicAdd(&r.ct[b], (1ull << 32) | (unsigned long long)n_inst);
    return make_uint2(b, (uint32_t)(old >> 32));
}
__device__ __forceinline__ void depth_register_end(const DepthReg &r, uint32_t idx, uint32_t key, uint2 bucket_ticket,
                                                   uint32_t payload = 0u)
{
    if (r.ct == nullptr) return;   // kernel-uniform
    uint32_t m0 = 0u, m1 = 0u, m2 = 0u, m3 = 0u;
    if (key != DEPTH_CULLED_KEY) {
        r.bt[idx] = bucket_ticket;
        if (r.payload != nullptr) r.payload[idx] = payload;
        if (key >> 31) { m2 = key; m3 = ~key; }
        else { m0 = key; m1 = ~key; }
    }
    __shared__ uint32_t dr_sm[4][4];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        m0 = max(m0, (uint32_t)__shfl_xor(m0, d));
        m1 = max(m1, (uint32_t)__shfl_xor(m1, d));
        m2 = max(m2, (uint32_t)__shfl_xor(m2, d));
        m3 = max(m3, (uint32_t)__shfl_xor(m3, d));
    }
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        dr_sm[0][w] = m0; dr_sm[1][w] = m1; dr_sm[2][w] = m2; dr_sm[3][w] = m3;
    }
    __syncthreads();
    if (threadIdx.x < 4)
        r.wgmm[blockIdx.x * 4u + threadIdx.x] = max(max(dr_sm[threadIdx.x][0], dr_sm[threadIdx.x][1]),
                                                    max(dr_sm[threadIdx.x][2], dr_sm[threadIdx.x][3]));
}
// host side of the fast path (depth_order.hip).  Words the host reads back after a forward pass, at the start of the
// depth order's temp storage (zeroed by depth_order_prepare):
enum { DW_TOTAL = 0, DW_OVERFLOW, DW_USER, DW_PMAX, DW_PNMAX, DW_NMAX, DW_NNMAX, DW_NVIS, DW_COUNT };
uint32_t *depth_order_words(void *temp, size_t P);
uint4 *depth_order_slots(void *temp, size_t P);   // [P] 16-byte records of the hinted path; also the small-grid survivor list
// which = 0 rasterizer, 1 voxelizer: separate hint histories.  Returns false when there is no usable hint for P keys (first
// call, P changed, hints switched off): the caller then runs the un-hinted path.
bool depth_hint_lookup(int which, size_t P, DepthHint *out);
void depth_hint_update(int which, size_t P, const uint32_t words[DW_COUNT], bool overflowed);
DepthReg depth_order_reg(void *temp, size_t P, const DepthHint &h, bool with_rects = false);
// with_rects: 16-byte records in (key, id) order, {id, inclusive instance offset, rectangle, 0} for j < nvis, left by
// depth_order_fast_finish(..., rects = true) -- everything the emission kernel needs in one coalesced load
const uint4 *depth_order_sorted_records(void *temp, size_t P);
// ... and, per TILE_SORT_GRANULE instances, the sorted position of the Gaussian that owns instance q * TILE_SORT_GRANULE
// (entries q < *cap only): lets a kernel that works by OUTPUT range find its Gaussians without a search
const uint32_t *depth_order_granule_owners(void *temp, size_t P, uint32_t *cap);
// after the producer kernel: dual prefix sum (-> DW_TOTAL, DW_NVIS, DW_OVERFLOW, key extrema are final after this launch
// pair: its last workgroup posts all DW_COUNT words + seq to the host mailbox, see host_mailbox_arm) ...
int depth_order_fast_scan(void *temp, size_t P, uint32_t producer_workgroups, uint32_t *mailbox, uint32_t seq, hipStream_t s);
// ... then placement + ranking: order[j] (j < nvis) = ids in (key, id) order, offsets[j] = inclusive instance offsets
int depth_order_fast_finish(void *temp, size_t P, const uint32_t *keys, const uint32_t *n_inst, uint32_t *order,
                            uint32_t *offsets, hipStream_t s, bool rects = false);
struct WorkListOut;
bool sort_is_single_pass(int end_bit);
// single-pass (<= 12 key bits) stable sort of instances by tile: ids_out[pos] = ids[index], inv_out[index] = pos,
// *counts_out = per-tile instance counts (device pointer into temp)
int sort_by_tile_single_pass(void *temp, size_t temp_bytes, const uint32_t *tiles, const uint32_t *ids, uint32_t *ids_out,
                             uint32_t *inv_out, size_t n, int end_bit, const uint32_t **counts_out, hipStream_t s,
                             const struct WorkListOut *work_out = nullptr /* also build tile ranges + work list */,
                             bool hist_ready = false /* the per-tile histograms are in place (tile_sort_plan) */);
// layout of the single-pass tile sort's per-tile digit histograms, for a key producer that builds them itself
struct TileSortPlan {
    uint32_t tile_keys;   // keys per sort tile (a multiple of TILE_SORT_GRANULE)
    uint32_t ntiles;
    int bits;             // radix = 1 << bits digit values (the whole key)
    uint32_t *H;          // [ntiles][1 << bits]
    uint32_t *skip;       // skip[0] must be set to 0
};
constexpr uint32_t TILE_SORT_GRANULE = 512;
bool tile_sort_plan(void *temp, size_t temp_bytes, size_t n, int end_bit, TileSortPlan *out);
// tiles[k] = t for k in ranges[t] (the backward's per-instance tile id when the sort did not scatter the keys);
int fill_tiles_from_ranges(const uint2 *ranges, size_t T, uint32_t *tiles, hipStream_t s);
size_t scan_gather_temp_bytes(int P);
int inclusive_scan_gather_u32(void *temp, size_t temp_bytes, const uint32_t *in, const uint32_t *order, uint32_t *out,
                              int P, hipStream_t s, uint32_t *total_out = nullptr /* device word receiving out[P-1] */);
// one small device->host read (n <= 16 words) through pinned memory + busy-wait on an event.  begin enqueues the copy on the
// stream, wait spins until it has landed: work enqueued between the two keeps the GPU busy while the host waits.
int current_device_slot();   // index of the current HIP device into small per-device host tables (binning.hip)
int device_cu_count();       // compute units of the current device (cached per device)
constexpr int R2_MAX_DEVICES = 64;
// > 64 KB of dynamic LDS for `kernel` on the current device, asked for once per device; state = the call site's
// static signed char [R2_MAX_DEVICES] table (zero-initialised)
bool allow_dynamic_lds(const void *kernel, int bytes, signed char *state);
size_t device_lds_optin_bytes();   // the device's real per-workgroup LDS limit (sharedMemPerBlockOptin)
int read_host_words_begin(const uint32_t *dev_words, int n, hipStream_t s);
int read_host_words_wait(uint32_t *out, int n);
int read_host_words(const uint32_t *dev_words, uint32_t *out, int n, hipStream_t s);
// zero-copy variant: a kernel stores the words and then `seq` (release, system scope) at mailbox[15]; the host spins on it
// host-side timing of a forward pass around its synchronisation point (r2_profile_host)
void host_mark_forward_begin();
void host_mark_wait_begin();
void host_mark_wait_end();
void host_mark_forward_end();
int host_mailbox_arm(uint32_t **mailbox /* device-visible pinned host memory, 16 words */, uint32_t *seq);
int host_mailbox_wait(uint32_t seq, uint32_t *out, int n, hipStream_t s);
void host_words_release();   // the calling thread's pinned words (r2_thread_release; also at thread exit)
// deferred num_rendered (binning.hip): a forward that does not wait posts its control words to a pool slot and returns a token
// (>= DEFER_TOKEN_FLAG, still a non-negative int); the backward resolves it.  -1 from acquire: no slot, wait as usual
constexpr int DEFER_TOKEN_FLAG = 0x40000000;
int defer_acquire(uint32_t **mailbox, uint32_t *seq, uint32_t cap);
int defer_resolve(int token, uint32_t *out, int n, uint32_t *cap, hipStream_t s, bool release);
bool defer_peek(int token, uint32_t *out, int n);
// tile ranges of the sorted list + point_list[k] = vals_unsorted[perm[k]] in the same pass
int tile_ranges(const uint32_t *tiles_sorted, const uint32_t *perm, const uint32_t *vals_unsorted, uint32_t *point_list,
                size_t R, uint2 *ranges, size_t T, hipStream_t s, bool ranges_zeroed = false);
uint32_t higher_msb(uint32_t n);
// forward work list: tile t owns work items [chunk_base[t], chunk_base[t+1]), one per `chunk` list entries
size_t build_work_temp_bytes(size_t T);
// work_tile[w] = {tile, first instance, one past the last instance, work items of that tile}
void launch_build_work(const uint2 *ranges, uint32_t T, uint32_t chunk, uint32_t *chunk_base, uint4 *work_tile,
                       void *temp /* build_work_temp_bytes(T) bytes, may be null for T <= 4096 */, hipStream_t s,
                       uint32_t min_len = 0 /* tiles with fewer instances get no work item */);
// many tiles (> 4096), for a caller whose kernels have summed the work items per block of build_work_block_tiles() consecutive tiles
// themselves (partial[], in the temp storage): one launch instead of two
uint32_t build_work_block_tiles();
void launch_build_work_from_partials(const uint2 *ranges, uint32_t T, uint32_t chunk, uint32_t *chunk_base, uint4 *work_tile,
                                     const uint32_t *partial, hipStream_t s, uint32_t min_len = 0);
// same, but the tile ranges themselves are derived from the per-tile instance counts of a single-pass tile sort
void launch_ranges_and_work(const uint32_t *tile_counts, uint32_t T, uint32_t chunk, uint2 *ranges, uint32_t *chunk_base,
                            uint4 *work_tile, hipStream_t s, uint32_t min_len = 0);

// Tile ranges + forward work list from the per-tile instance counts, by ONE workgroup of NT threads (T <= 4096):
// ranges[t] = exclusive scan of counts (empty tiles keep (0,0) like the reference's memset), chunk_base[t] = exclusive
// scan of ceil(count / chunk), one 16-byte work descriptor {tile, first, last, items of the tile} per work item.
// Called by the standalone kernel in binning.hip and by block 0 of the single-pass tile sort's downsweep.
struct WorkListOut {
    uint2 *ranges;
    uint32_t *chunk_base;
    uint4 *work;
    uint32_t T, chunk;
    // optional (rasterizer, fused combine): arrival counters to zero ([4 T], 16-byte aligned), and EMPTY tiles appended to the work list as
    // items {tile, 0, 0, 0} behind the real ones (somebody has to write their zeros); chunk_base[T + 1] = real + empty items
    uint32_t *tile_done;
    // optional (voxelizer): tiles with fewer than min_len instances get NO work item (a light kernel renders them)
    uint32_t min_len;
    // optional: capacity of `work` in items (0 = as many as needed); items beyond it are dropped (the caller checks the total)
    uint32_t work_cap;
    // optional (rasterizer, with tile_done; round 6): the work list LONGEST FIRST -- every full item (chunk entries) in tile order,
    // then the tiles' remainders by descending length class, then the empty tiles -- so that the kernel's last round of waves is
    // made of its shortest items.  chunk_base[t] stays the tile-order prefix (the index space of a tile's partial sums: item
    // chunk_base[t] + j is the tile's j-th chunk); only the ORDER of `work` changes, so a consumer must not take an item's
    // position in `work` for that index.
    uint32_t longest_first;
};
constexpr uint32_t WORK_CLASSES = 16;
template <int NT>
__device__ __forceinline__ void ranges_and_work_block(const uint32_t *__restrict__ counts, const WorkListOut wo)
{
    __shared__ uint32_t rw_wsum[NT / 64], rw_wsum2[NT / 64], rw_wsum3[NT / 64];
    __shared__ uint32_t rw_carry, rw_carry2, rw_carry3, rw_cls[WORK_CLASSES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool lf = wo.longest_first != 0u && wo.tile_done != nullptr;   // (kernel-uniform)
    if (tid == 0) { rw_carry = 0; rw_carry2 = 0; rw_carry3 = 0; }
    if (tid < (int)WORK_CLASSES) rw_cls[tid] = 0u;
    __syncthreads();
    // length class of a remainder of `rem` entries (0 < rem < chunk): 0 = the longest
    auto cls_of = [&](uint32_t rem) { return WORK_CLASSES - 1u - min(WORK_CLASSES - 1u, (rem * WORK_CLASSES) / (wo.chunk & ~WORK_CHUNK_ADAPT)); };
    for (uint32_t base = 0; base < wo.T; base += NT) {
        const uint32_t t = base + tid;
        const uint32_t c = t < wo.T ? counts[t] : 0u;
        const uint32_t ch = work_tile_chunk(wo.chunk, c);   // (the adaptive rule and the longest-first order do not meet: voxelizer / rasterizer)
        const uint32_t nw = c < wo.min_len ? 0u : (c + ch - 1) / ch;
        const uint32_t nfull = lf ? c / ch : nw;   // items placed in this sweep (longest first: the full ones)
        uint32_t incl = c, incl2 = nw, incl3 = nfull;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d), up2 = __shfl_up(incl2, d), up3 = __shfl_up(incl3, d);
            if (lane >= d) { incl += up; incl2 += up2; incl3 += up3; }
        }
        if (lane == 63) { rw_wsum[wave] = incl; rw_wsum2[wave] = incl2; rw_wsum3[wave] = incl3; }
        __syncthreads();
        uint32_t woff = 0, woff2 = 0, woff3 = 0;
        for (int w = 0; w < wave; ++w) { woff += rw_wsum[w]; woff2 += rw_wsum2[w]; woff3 += rw_wsum3[w]; }
        const uint32_t start = rw_carry + woff + incl - c, wstart = rw_carry2 + woff2 + incl2 - nw;
        const uint32_t fstart = lf ? rw_carry3 + woff3 + incl3 - nfull : wstart;
        if (t < wo.T) {
            wo.ranges[t] = c ? make_uint2(start, start + c) : make_uint2(0u, 0u);
            wo.chunk_base[t] = wstart;
            for (uint32_t j = 0; j < nfull; ++j)
                if (!wo.work_cap || fstart + j < wo.work_cap)
                    wo.work[fstart + j] = make_uint4(t, start + j * ch, min(start + c, start + (j + 1) * ch), nw);
            if (lf && nw != nfull) atomicAdd(&rw_cls[cls_of(c - nfull * ch)], 1u);
        }
        __syncthreads();
        if (tid == NT - 1) { rw_carry = start + c; rw_carry2 = wstart + nw; rw_carry3 = fstart + nfull; }
        __syncthreads();
    }
    if (tid == 0) wo.chunk_base[wo.T] = rw_carry2;
    if (wo.tile_done == nullptr) return;
    // second sweep: zero the arrival counters, place the remainders (longest first), append the empty tiles
    const uint32_t nreal = rw_carry2;
    __syncthreads();
    if (tid == 0) {
        rw_carry = 0;
        uint32_t run = rw_carry3;   // the remainders follow the full items, class after class
        for (uint32_t k = 0; k < WORK_CLASSES; ++k) { const uint32_t n = rw_cls[k]; rw_cls[k] = run; run += n; }
    }
    __syncthreads();
    for (uint32_t base = 0; base < wo.T; base += NT) {
        const uint32_t t = base + tid;
        const uint32_t empty = (t < wo.T && counts[t] == 0u) ? 1u : 0u;
        if (lf && t < wo.T) {
            const uint32_t c = counts[t], nfull = c / wo.chunk, rem = c - nfull * wo.chunk;
            if (rem != 0u && c >= wo.min_len) {
                const uint32_t pos = atomicAdd(&rw_cls[cls_of(rem)], 1u);   // (the order inside a class is immaterial)
                const uint32_t start = wo.ranges[t].x;                      // written by this thread in the first sweep
                if (!wo.work_cap || pos < wo.work_cap)
                    wo.work[pos] = make_uint4(t, start + nfull * wo.chunk, start + c, nfull + 1u);
            }
        }
        if (t < wo.T) reinterpret_cast<uint4 *>(wo.tile_done)[t] = make_uint4(0u, 0u, 0u, 0u);   // one counter per 8x8 block
        uint32_t incl = empty;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) rw_wsum[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += rw_wsum[w];
        const uint32_t pos = rw_carry + woff + incl - empty;
        if (empty) wo.work[nreal + pos] = make_uint4(t, 0u, 0u, 0u);
        __syncthreads();
        if (tid == NT - 1) rw_carry = pos + empty;
        __syncthreads();
    }
    if (tid == 0) wo.chunk_base[wo.T + 1] = nreal + rw_carry;
}

// XCD-aware remap of a linear block id: consecutive work items (neighbouring tiles / list chunks,
// which share Gaussian records) stay on one XCD's L2 instead of being dealt round-robin over 8.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n)
{
    const uint32_t per = (n + 7u) >> 3;
    const uint32_t w = (b & 7u) * per + (b >> 3);
    return w;   // may be >= n for the ragged tail: caller checks
}

}  // namespace r2
