// depth_order.hip -- Gaussian ids in (depth key, id) order: the first half of the two-stage exact sort (see binning.hip).
//
// A general 32-bit LSD radix sort needs 2-3 digit passes of 3 kernels each and, with wide digits, pays for fully
// scattered 4-byte stores (radix_sort.hip); for P ~ 1e5..1e6 keys that is ~90 us of launch latency.  The depth keys are
// float bit patterns spread over a narrow range, which a ONE-LEVEL bucket sort exploits:
//   1. min / max of the visible keys;
//   2. bucket = (key - min) >> shift with ~P buckets (monotone in the key), counted with one global atomic per key;
//   3. prefix sum of the bucket counts;
//   4. each key takes a slot of its bucket (atomic ticket: order inside the bucket is arbitrary at this point);
//   5. every slot ranks its key among the bucket's (key, id) pairs -- buckets hold ~1 key on average -- and writes the id
//      to its final position.  Culled Gaussians (key 0xFFFFFFFF) go to the tail.
// The result is exactly the stable sort by key.  If some bucket is too full for step 5 (many identical depths) a flag
// is raised and the caller re-sorts with the radix sort; the flag is read at the host synchronisation the forward
// pass has anyway.
#include "r2_common.hpp"
#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstdlib>
#include <cmath>

R2_TS_DEFINE(order)

namespace r2 {

namespace {

constexpr uint32_t CULLED_KEY = DEPTH_CULLED_KEY;
constexpr uint32_t MAX_BUCKET = 256;  // a fuller bucket raises the fallback flag (ranking costs one pass over the bucket per key)

struct Ctrl {           // zeroed before every use (depth_order_prepare); the first DW_COUNT words are what the host reads back
    uint32_t total;     // DW_TOTAL    number of instances (sum of the per-Gaussian instance counts)
    uint32_t overflow;  // DW_OVERFLOW a bucket was too full: the order is invalid, fall back
    uint32_t user;      // DW_USER     a word the caller's producer kernel may set
    uint32_t pmax;      // DW_PMAX..   max / (complement of the) min of the keys with a clear sign bit (positive floats) ...
    uint32_t pnmax;
    uint32_t nmax;      //             ... and of the keys with the sign bit set (negative floats: bits grow with |value|)
    uint32_t nnmax;
    uint32_t nvis;      // DW_NVIS     visible keys (hinted path)
    uint32_t nculled;   // ticket counter of culled Gaussians (un-hinted path)
    uint32_t pad[7];
};
static_assert(offsetof(Ctrl, nvis) == DW_NVIS * sizeof(uint32_t) && offsetof(Ctrl, pmax) == DW_PMAX * sizeof(uint32_t), "host words");

// Bucket of a key: monotone non-decreasing in the key's UNSIGNED bit pattern (the sort order), and roughly uniform in
// occupancy for keys that are float bit patterns: positive floats come first (their bits grow with the value), then the
// negative ones (bits grow with |value|); inside each class the bucket is linear in the float VALUE between the class's
// min and max -- depths are spread evenly in value, not in bit pattern (half of all floats in (0,1) lie in [0.5,1)).
// NaN / inf land in the last bucket of their class (fminf), which keeps the map monotone.
__device__ __forceinline__ uint32_t bucket_of(uint32_t key, const Ctrl *__restrict__ c, int log_nb)
{
    const uint32_t nb = 1u << log_nb;
    const bool has_neg = c->nmax != 0u, has_pos = c->pmax != 0u || c->pnmax != 0u;
    const uint32_t nbp = has_neg ? (has_pos ? nb >> 1 : 0u) : nb;   // buckets given to the positive class
    const bool neg = (key >> 31) != 0u;
    const float v = __uint_as_float(key & 0x7FFFFFFFu);
    const float lo = __uint_as_float((neg ? ~c->nnmax : ~c->pnmax) & 0x7FFFFFFFu);
    const float hi = __uint_as_float((neg ? c->nmax : c->pmax) & 0x7FFFFFFFu);
    const uint32_t cnt = neg ? nb - nbp : nbp;
    const float scale = hi > lo ? (float)(cnt - 1u) / (hi - lo) : 0.f;
    const uint32_t b = (uint32_t)fminf(fmaxf((v - lo) * scale, 0.f), (float)(cnt - 1u));
    return (neg ? nbp : 0u) + b;
}

// few, fat workgroups and ONE atomic group per workgroup: same-address atomics retire at only ~90 per microsecond
__global__ void __launch_bounds__(1024) minmax_kernel(const uint32_t *__restrict__ keys, uint32_t n, Ctrl *__restrict__ c)
{
    __shared__ uint32_t sm[4][16];
    uint32_t m[4] = { 0u, 0u, 0u, 0u };   // pmax, pnmax, nmax, nnmax
    for (uint32_t i = blockIdx.x * 1024u + threadIdx.x; i < n; i += gridDim.x * 1024u) {
        const uint32_t k = keys[i];
        if (k == CULLED_KEY) continue;
        if (k >> 31) { m[2] = max(m[2], k); m[3] = max(m[3], ~k); }
        else { m[0] = max(m[0], k); m[1] = max(m[1], ~k); }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m[q] = max(m[q], (uint32_t)__shfl_xor(m[q], d));
        if ((threadIdx.x & 63) == 0) sm[q][threadIdx.x >> 6] = m[q];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        uint32_t r = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) r = max(r, sm[threadIdx.x][w]);
        if (r) atomicMax(&c->pmax + threadIdx.x, r);
    }
}

__global__ void __launch_bounds__(256) bucket_count_kernel(const uint32_t *__restrict__ keys, uint32_t n, const Ctrl *__restrict__ c,
                                                           int log_nb, uint32_t *__restrict__ counts)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = keys[i];
    if (k == CULLED_KEY) return;
    atomicAdd(&counts[bucket_of(k, c, log_nb)], 1u);
}

__global__ void __launch_bounds__(256) bucket_place_kernel(const uint32_t *__restrict__ keys, uint32_t n, Ctrl *__restrict__ c,
                                                           int log_nb, uint32_t *__restrict__ counts,
                                                           const uint32_t *__restrict__ incl, uint32_t *__restrict__ slot_key,
                                                           uint32_t *__restrict__ slot_id, uint32_t *__restrict__ order)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t k = i < n ? keys[i] : 0u;
    // culled Gaussians go to the tail in any order (they emit nothing); ONE ticket atomic per wave -- same-address atomics
    // retire at ~90 per microsecond, a scene with 100 k culled Gaussians would otherwise spend a millisecond here
    const bool culled = i < n && k == CULLED_KEY;
    const unsigned long long cm = __ballot(culled);
    if (cm) {
        const int lane = threadIdx.x & 63;
        uint32_t base = 0;
        if (lane == __ffsll((long long)cm) - 1) base = atomicAdd(&c->nculled, (uint32_t)__popcll(cm));
        base = __shfl(base, __ffsll((long long)cm) - 1);
        if (culled) order[n - 1u - (base + (uint32_t)__popcll(cm & ((1ull << lane) - 1ull)))] = i;
    }
    if (i >= n || culled) return;
    const uint32_t b = bucket_of(k, c, log_nb);
    const uint32_t v = atomicSub(&counts[b], 1u);          // v in [1, count]: a unique slot inside the bucket
    const uint32_t pos = incl[b] - v;
    slot_key[pos] = k;
    slot_id[pos] = i;
}

__global__ void __launch_bounds__(256) bucket_rank_kernel(uint32_t nb, Ctrl *__restrict__ c, int log_nb,
                                                          const uint32_t *__restrict__ incl, const uint32_t *__restrict__ slot_key,
                                                          const uint32_t *__restrict__ slot_id, uint32_t *__restrict__ order)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    const uint32_t nvis = incl[nb - 1];
    uint32_t fp = 0xFFFFFFFFu, id = 0;
    if (p < nvis) {
        const uint32_t k = slot_key[p];
        id = slot_id[p];
        const uint32_t b = bucket_of(k, c, log_nb);
        const uint32_t beg = b ? incl[b - 1] : 0u, end = incl[b];
        const uint32_t m = end - beg;
        if (m == 1) {
            fp = p;
        } else if (m > MAX_BUCKET) {   // invalid result, flagged; still leave a valid permutation behind
            c->overflow = 1u;
            fp = p;
        } else {
            uint32_t rank = 0;
            for (uint32_t q = beg; q < end; ++q) {
                const uint32_t kq = slot_key[q], iq = slot_id[q];
                rank += (kq < k || (kq == k && iq < id)) ? 1u : 0u;
            }
            fp = beg + rank;
        }
        order[fp] = id;
    }
}

// ------------------------------------------------------------------------------------------ hinted fast path
// (see r2_common.hpp: the producer kernel has filled counts / tsum / bt / wgmm through depth_register)
constexpr int S2_THREADS = 1024;
constexpr int S2_IPT = 4;
constexpr int S2_TILE = S2_THREADS * S2_IPT;

__device__ __forceinline__ uint2 block_reduce2_1024(uint2 v, uint2 *sh /* [16] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        v.x += __shfl_xor(v.x, d);
        v.y += __shfl_xor(v.y, d);
    }
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    uint2 t = make_uint2(0u, 0u);
#pragma unroll
    for (int w = 0; w < S2_THREADS / 64; ++w) { t.x += sh[w].x; t.y += sh[w].y; }
    __syncthreads();
    return t;
}

// (Measured and left out, round 2: ONE kernel for the dual prefix sum -- every workgroup publishes its two sums as a 64-bit word
// with agent-scope atomics and waits only for the words of the workgroups before it, no chain.  Parity green; the cross-XCD
// publish / wait round trip costs what the kernel boundary costs: the place kernel started 0.9 us LATER.)
// per-4096-bucket partial sums of both arrays; flags over-full buckets; the last workgroup + 1 folds the producer's
// per-workgroup key extrema into the control block (the next call's hint)
__global__ void __launch_bounds__(S2_THREADS) scan2_reduce_kernel(const uint2 *__restrict__ ct /* {instances, keys} */,
                                                                  uint32_t nb, uint2 *__restrict__ partial, Ctrl *__restrict__ c,
                                                                  const uint32_t *__restrict__ wgmm, uint32_t nwg)
{
    __shared__ uint2 sh[S2_THREADS / 64];
    R2_TS_AT(order, 2);
    if (blockIdx.x == gridDim.x - 1) {   // extra workgroup: extrema
        uint32_t m[4] = { 0u, 0u, 0u, 0u };
        for (uint32_t g = threadIdx.x; g < nwg; g += S2_THREADS) {
            const uint4 v = reinterpret_cast<const uint4 *>(wgmm)[g];
            m[0] = max(m[0], v.x); m[1] = max(m[1], v.y); m[2] = max(m[2], v.z); m[3] = max(m[3], v.w);
        }
        __shared__ uint32_t smm[4][S2_THREADS / 64];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) m[q] = max(m[q], (uint32_t)__shfl_xor(m[q], d));
            if ((threadIdx.x & 63) == 0) smm[q][threadIdx.x >> 6] = m[q];
        }
        __syncthreads();
        if (threadIdx.x < 4) {
            uint32_t r = 0;
#pragma unroll
            for (int w = 0; w < S2_THREADS / 64; ++w) r = max(r, smm[threadIdx.x][w]);
            (&c->pmax)[threadIdx.x] = r;
        }
        return;
    }
    const uint32_t base = blockIdx.x * S2_TILE + threadIdx.x * S2_IPT;
    uint2 v = make_uint2(0u, 0u);
    bool over = false;
    if (base + S2_IPT <= nb) {
        const uint4 q0 = *reinterpret_cast<const uint4 *>(ct + base), q1 = *reinterpret_cast<const uint4 *>(ct + base + 2);
        v.x = q0.y + q0.w + q1.y + q1.w;   // keys
        v.y = q0.x + q0.z + q1.x + q1.z;   // instances
        over = max(max(q0.y, q0.w), max(q1.y, q1.w)) > MAX_BUCKET;
    }
    if (over) c->overflow = 1u;   // benign race: everybody stores 1
    const uint2 t = block_reduce2_1024(v, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
    R2_TS_AT(order, 3);
}

// inclusive prefix sums of both arrays; the last element's sums are the number of visible keys and of instances
__global__ void __launch_bounds__(S2_THREADS) scan2_apply_kernel(const uint2 *__restrict__ ct, uint32_t nb,
                                                                 const uint2 *__restrict__ partial,
                                                                 uint32_t *__restrict__ incl_c, uint32_t *__restrict__ incl_t,
                                                                 Ctrl *__restrict__ c, uint32_t *__restrict__ mailbox, uint32_t seq)
{
    __shared__ uint2 sh[S2_THREADS / 64];
    __shared__ uint2 wsum[S2_THREADS / 64];
    R2_TS_AT(order, 4);
    uint2 pre = make_uint2(0u, 0u);
    for (uint32_t g = threadIdx.x; g < blockIdx.x; g += S2_THREADS) {
        const uint2 q = partial[g];
        pre.x += q.x; pre.y += q.y;
    }
    const uint2 tile_base = block_reduce2_1024(pre, sh);
    const uint32_t base = blockIdx.x * S2_TILE + threadIdx.x * S2_IPT;
    const uint4 q0 = *reinterpret_cast<const uint4 *>(ct + base), q1 = *reinterpret_cast<const uint4 *>(ct + base + 2);
    const uint4 cc = make_uint4(q0.y, q0.w, q1.y, q1.w), tt = make_uint4(q0.x, q0.z, q1.x, q1.z);
    const uint32_t sc = cc.x + cc.y + cc.z + cc.w, st = tt.x + tt.y + tt.z + tt.w;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t ic = sc, it = st;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t uc = __shfl_up(ic, d), ut = __shfl_up(it, d);
        if (lane >= d) { ic += uc; it += ut; }
    }
    if (lane == 63) wsum[wave] = make_uint2(ic, it);
    __syncthreads();
    uint32_t rc = tile_base.x + ic - sc, rt = tile_base.y + it - st;
    for (int w = 0; w < wave; ++w) { rc += wsum[w].x; rt += wsum[w].y; }
    uint4 oc, ot;
    oc.x = rc + cc.x; oc.y = oc.x + cc.y; oc.z = oc.y + cc.z; oc.w = oc.z + cc.w;
    ot.x = rt + tt.x; ot.y = ot.x + tt.y; ot.z = ot.y + tt.z; ot.w = ot.z + tt.w;
    *reinterpret_cast<uint4 *>(incl_c + base) = oc;
    *reinterpret_cast<uint4 *>(incl_t + base) = ot;
    if (base + S2_IPT == nb) {
        // the instance counts are scanned in 32 bits; their total in 64, so that a sum beyond the 31-bit num_rendered of
        // the API is reported (as an out-of-range total the host rejects) instead of wrapping silently
        unsigned long long total64 = 0;
        for (uint32_t g = 0; g < gridDim.x; ++g) total64 += partial[g].y;
        if (total64 > 0x7FFFFFFFull) ot.w = 0xFFFFFFFFu;
        c->nvis = oc.w;
        c->total = ot.w;
        if (mailbox) {   // post the host-read words (the others were written by earlier kernels) straight to pinned host memory
            mailbox[DW_TOTAL] = ot.w;
            mailbox[DW_OVERFLOW] = c->overflow;
            mailbox[DW_USER] = c->user;
            mailbox[DW_PMAX] = c->pmax;
            mailbox[DW_PNMAX] = c->pnmax;
            mailbox[DW_NMAX] = c->nmax;
            mailbox[DW_NNMAX] = c->nnmax;
            mailbox[DW_NVIS] = oc.w;
            __hip_atomic_store(&mailbox[15], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    R2_TS_AT(order, 5);
}

__global__ void __launch_bounds__(256) zero_kernel(uint4 *__restrict__ p, size_t n16)
{
    R2_TS_AT(order, 0);
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256u) p[i] = make_uint4(0u, 0u, 0u, 0u);
    R2_TS_AT(order, 1);
}

// every visible key takes its slot: bucket base + ticket.  One 16-byte record {key, id, instances, bucket} per slot
// (RECT: n_inst = the producer's payload words, tile rectangles that encode the instance count -- depth_rect_count).
__global__ void __launch_bounds__(256) fast_place_kernel(uint32_t n, const uint32_t *__restrict__ keys, const uint32_t *__restrict__ n_inst,
                                                         const uint2 *__restrict__ bt, const uint32_t *__restrict__ incl_c,
                                                         uint4 *__restrict__ slot, bool rect)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    R2_TS_AT(order, 6);
    if (i >= n) return;
    const uint32_t k = keys[i], m = n_inst[i];
    if (k == CULLED_KEY || (!rect && m == 0u)) return;   // (a rectangle word may be 0: one tile at the origin)
    const uint2 b = bt[i];
    const uint32_t beg = b.x ? incl_c[b.x - 1u] : 0u;
    slot[beg + b.y] = make_uint4(k, i, m, b.x);
    R2_TS_AT(order, 7);
}

// every slot ranks its key among the (key, id) pairs of its bucket, which also gives the instances emitted before it
// (Measured and left out, round 2: a lane per BUCKET instead of per slot -- bucket bounds first, the wave's slot range
// staged in LDS, two dependent round trips instead of three, 1.4 us for the median workgroup -- but the clamped end buckets of
// a hinted range routinely hold up to MAX_BUCKET keys, and the lane that owns one ranks n^2 pairs alone: 90-140 us for that
// workgroup.  A lane per slot spreads exactly those buckets over many lanes.
// Round 3: the members of a bucket are neighbouring LANES (consecutive slots), so the placement kernel stored the bucket's first
// slot and its instance base in the record and this kernel read the other members out of the wave's registers (ds_bpermute; only
// buckets that cross a wave walked memory) -- one memory round trip instead of three, indices identical: place + rank 19.5 -> 24.1 us.
// The per-member shuffles (three per member, serial) cost more than the L1-resident loads they replace.)
template <bool RECT>
__global__ void __launch_bounds__(256) fast_rank_kernel(Ctrl *__restrict__ c, const uint4 *__restrict__ slot,
                                                        const uint32_t *__restrict__ incl_c, const uint32_t *__restrict__ incl_t,
                                                        uint32_t *__restrict__ order, uint32_t *__restrict__ offsets,
                                                        uint4 *__restrict__ sorted, uint32_t *__restrict__ owners, uint32_t owners_cap)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    R2_TS_AT(order, 8);
    if (p >= c->nvis) return;
    const uint4 me = slot[p];
    const uint32_t b = me.w;
    const uint32_t beg = b ? incl_c[b - 1u] : 0u, end = incl_c[b];
    const uint32_t tbeg = b ? incl_t[b - 1u] : 0u;
    uint32_t rank = 0, before = 0;
    if (end - beg > MAX_BUCKET) {   // flagged by the scan already; leave a valid permutation behind
        order[p] = me.y;
        offsets[p] = 0u;
        return;
    }
    for (uint32_t q = beg; q < end; ++q) {
        if (q == p) continue;
        const uint4 o = slot[q];
        const bool lt = o.x < me.x || (o.x == me.x && o.y < me.y);
        rank += lt ? 1u : 0u;
        before += lt ? (RECT ? depth_rect_count(o.z) : o.z) : 0u;
    }
    const uint32_t incl = tbeg + before + (RECT ? depth_rect_count(me.z) : me.z);   // inclusive, like the scan of the un-hinted path
    order[beg + rank] = me.y;
    offsets[beg + rank] = incl;
    if (RECT) {
        sorted[beg + rank] = make_uint4(me.y, incl, me.z, 0u);
        // owner of every TILE_SORT_GRANULE-th instance (a Gaussian emits ~4 instances: one lane in a hundred writes one entry)
        const uint32_t excl = incl - depth_rect_count(me.z);
        for (uint32_t q = (excl + TILE_SORT_GRANULE - 1u) / TILE_SORT_GRANULE; q * TILE_SORT_GRANULE < incl && q < owners_cap; ++q)
            owners[q] = beg + rank;
    }
    R2_TS_AT(order, 9);
}

struct Temp {
    Ctrl *ctrl;
    uint32_t *counts;     // [nb]   un-hinted path (ctrl and counts / ct are zeroed by one launch)
    unsigned long long *ct; // [nb] hinted path: (keys << 32) | instances per bucket; shares its memory with counts
    uint32_t *incl;       // [nb]
    uint32_t *incl_t;     // [nb]   hinted path
    uint32_t *slot_key;   // [P]    un-hinted path
    uint32_t *slot_id;    // [P]
    uint4 *slot;          // [P]    hinted path (aliases nothing: the fallback may run after a failed hinted attempt)
    uint2 *bt;            // [P]
    uint32_t *wgmm;       // [4 * (P/256 + 1)]
    uint2 *partial2;      // [nb/4096 + 1]
    uint32_t *payload;    // [P]    hinted path, optional (DepthReg::payload)
    uint4 *sorted;        // [P]    hinted path, optional: sorted records
    uint32_t *owners;     // [P + 4096] ... and the owners of every TILE_SORT_GRANULE-th instance
    char *scan_temp;
    size_t scan_bytes, zero_bytes, bytes;
    static Temp carve(char *chunk, size_t P, size_t nb)
    {
        Temp t;
        Bump b(chunk);
        t.ctrl = b.take<Ctrl>(2);               // 128 bytes: keeps counts on the next 128-byte boundary
        t.ct = b.take<unsigned long long>(nb);
        t.counts = reinterpret_cast<uint32_t *>(t.ct);
        t.zero_bytes = b.offset_of_next();
        t.incl = b.take<uint32_t>(nb);
        t.incl_t = b.take<uint32_t>(nb);
        t.slot_key = b.take<uint32_t>(P);
        t.slot_id = b.take<uint32_t>(P);
        t.slot = b.take<uint4>(P);
        t.bt = b.take<uint2>(P);
        t.wgmm = b.take<uint32_t>(4 * (P / 256 + 2));
        t.partial2 = b.take<uint2>(nb / S2_TILE + 2);
        t.payload = b.take<uint32_t>(P);
        t.sorted = b.take<uint4>(P);
        t.owners = b.take<uint32_t>(P + 4096);
        t.scan_bytes = scan_temp_bytes((int)nb);
        t.scan_temp = b.take<char>(t.scan_bytes);
        t.bytes = b.total();
        return t;
    }
};

inline int log_buckets(size_t P)
{
    int l = 12;
    while (l < 22 && ((size_t)4 << l) < P) ++l;   // 2-4 keys per bucket, 2^12 .. 2^22 buckets (a multiple of the scan tile):
    return l;                                     // the counters have to be zeroed and scanned on every call
}

}  // namespace

size_t depth_order_temp_bytes(size_t P) { return Temp::carve(nullptr, P, (size_t)1 << log_buckets(P)).bytes; }

// zeroes the control block and the counters; called BEFORE the kernel that produces the keys, which may then set the user
// word (and, on the hinted path, registers its keys)
int depth_order_prepare(void *temp, size_t temp_bytes, size_t P, hipStream_t s)
{
    if (P == 0) return 0;
    const Temp t = Temp::carve(reinterpret_cast<char *>(temp), P, (size_t)1 << log_buckets(P));
    if (t.bytes > temp_bytes) {
        set_error("depth_order_prepare: temp storage too small (%zu < %zu)", temp_bytes, t.bytes);
        return R2_ERR_INVALID;
    }
    // one launch (hipMemsetAsync splits a fill of this size into two ~5 us kernels); zero_bytes is a multiple of 128
    const size_t n16 = t.zero_bytes / 16;
    zero_kernel<<<dim3((unsigned)std::min<size_t>((n16 + 255) / 256, 2048)), dim3(256), 0, s>>>(reinterpret_cast<uint4 *>(temp), n16);
    R2_HIP_TRY(hipGetLastError());
    return 0;
}
uint4 *depth_order_slots(void *temp, size_t P)
{
    return Temp::carve(reinterpret_cast<char *>(temp), P, (size_t)1 << log_buckets(P)).slot;
}
uint32_t *depth_order_words(void *temp, size_t P)
{
    return &Temp::carve(reinterpret_cast<char *>(temp), P, (size_t)1 << log_buckets(P)).ctrl->total;
}

// Un-hinted path.  order[P] = Gaussian ids sorted by (key, id); culled ids (key 0xFFFFFFFF) at the tail in arbitrary order.
// The control block's overflow word is set when the result is INVALID (fall back to the radix sort); it must be read
// after the stream has caught up.  depth_order_prepare must have been called on temp (and nothing but the producer kernel
// may have touched it since).
int depth_order_buckets(void *temp, size_t temp_bytes, const uint32_t *keys, uint32_t *order, size_t P, hipStream_t s)
{
    if (P == 0) return 0;
    const int log_nb = log_buckets(P);
    const size_t nb = (size_t)1 << log_nb;
    const Temp t = Temp::carve(reinterpret_cast<char *>(temp), P, nb);
    if (t.bytes > temp_bytes) {
        set_error("depth_order_buckets: temp storage too small (%zu < %zu)", temp_bytes, t.bytes);
        return R2_ERR_INVALID;
    }
    const unsigned grid = (unsigned)((P + 255) / 256);
    minmax_kernel<<<dim3(std::min((unsigned)((P + 4095) / 4096), 64u)), dim3(1024), 0, s>>>(keys, (uint32_t)P, t.ctrl);
    bucket_count_kernel<<<dim3(grid), dim3(256), 0, s>>>(keys, (uint32_t)P, t.ctrl, log_nb, t.counts);
    const int rc = inclusive_scan_u32(t.scan_temp, t.scan_bytes, t.counts, t.incl, (int)nb, s);
    if (rc) return rc;
    bucket_place_kernel<<<dim3(grid), dim3(256), 0, s>>>(keys, (uint32_t)P, t.ctrl, log_nb, t.counts, t.incl, t.slot_key,
                                                         t.slot_id, order);
    bucket_rank_kernel<<<dim3(grid), dim3(256), 0, s>>>((uint32_t)nb, t.ctrl, log_nb, t.incl, t.slot_key, t.slot_id, order);
    R2_HIP_TRY(hipGetLastError());
    return 0;
}

// ---- hinted path, host side
DepthReg depth_order_reg(void *temp, size_t P, const DepthHint &h, bool with_rects)
{
    const Temp t = Temp::carve(reinterpret_cast<char *>(temp), P, (size_t)1 << log_buckets(P));
    return DepthReg{t.ct, t.bt, t.wgmm, h, with_rects ? t.payload : nullptr};
}
const uint4 *depth_order_sorted_records(void *temp, size_t P)
{
    return Temp::carve(reinterpret_cast<char *>(temp), P, (size_t)1 << log_buckets(P)).sorted;
}

int depth_order_fast_scan(void *temp, size_t P, uint32_t producer_workgroups, uint32_t *mailbox, uint32_t seq, hipStream_t s)
{
    const size_t nb = (size_t)1 << log_buckets(P);
    const Temp t = Temp::carve(reinterpret_cast<char *>(temp), P, nb);
    const unsigned tiles = (unsigned)(nb / S2_TILE);
    scan2_reduce_kernel<<<dim3(tiles + 1), dim3(S2_THREADS), 0, s>>>(reinterpret_cast<const uint2 *>(t.ct), (uint32_t)nb, t.partial2, t.ctrl,
                                                                      t.wgmm, producer_workgroups);
    scan2_apply_kernel<<<dim3(tiles), dim3(S2_THREADS), 0, s>>>(reinterpret_cast<const uint2 *>(t.ct), (uint32_t)nb, t.partial2, t.incl, t.incl_t,
                                                                t.ctrl, mailbox, seq);
    R2_HIP_TRY(hipGetLastError());
    return 0;
}

const uint32_t *depth_order_granule_owners(void *temp, size_t P, uint32_t *cap)
{
    *cap = (uint32_t)(P + 4096);
    return Temp::carve(reinterpret_cast<char *>(temp), P, (size_t)1 << log_buckets(P)).owners;
}

int depth_order_fast_finish(void *temp, size_t P, const uint32_t *keys, const uint32_t *n_inst, uint32_t *order,
                            uint32_t *offsets, hipStream_t s, bool rects)
{
    const Temp t = Temp::carve(reinterpret_cast<char *>(temp), P, (size_t)1 << log_buckets(P));
    const unsigned grid = (unsigned)((P + 255) / 256);
    fast_place_kernel<<<dim3(grid), dim3(256), 0, s>>>((uint32_t)P, keys, rects ? t.payload : n_inst, t.bt, t.incl, t.slot, rects);
    if (rects) fast_rank_kernel<true><<<dim3(grid), dim3(256), 0, s>>>(t.ctrl, t.slot, t.incl, t.incl_t, order, offsets, t.sorted, t.owners,
                                                                       (uint32_t)(P + 4096));
    else fast_rank_kernel<false><<<dim3(grid), dim3(256), 0, s>>>(t.ctrl, t.slot, t.incl, t.incl_t, order, offsets, nullptr, nullptr, 0u);
    R2_HIP_TRY(hipGetLastError());
    return 0;
}

// ---- hint history (host): the key range seen by the previous call(s) with the same number of keys
namespace {
struct HintEntry {
    size_t P = 0;
    bool valid = false;
    bool has_pos = false, has_neg = false;
    float plo = 0.f, phi = 0.f, nlo = 0.f, nhi = 0.f;   // value ranges (negative class: |value|)
    unsigned age = 0;
};
thread_local HintEntry g_hints[2][4];
thread_local unsigned g_hint_clock[2] = { 0, 0 };
std::atomic<int> g_hint_mode{-1};   // -1: read R2_DEPTH_HINT from the environment on first use; 0 off; 1 on

bool hints_enabled()
{
    int m = g_hint_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        const char *e = getenv("R2_DEPTH_HINT");
        int want = (e && e[0] == '0') ? 0 : 1;
        // several threads may get here at once: the first to install its (identical) answer wins
        int expected = -1;
        g_hint_mode.compare_exchange_strong(expected, want, std::memory_order_relaxed);
        m = g_hint_mode.load(std::memory_order_relaxed);
    }
    return m == 1;
}
HintEntry *find_hint(int which, size_t P)
{
    for (auto &h : g_hints[which])
        if (h.valid && h.P == P) return &h;
    return nullptr;
}
}  // namespace

bool depth_hint_lookup(int which, size_t P, DepthHint *out)
{
    if (!hints_enabled() || which < 0 || which > 1) return false;
    const HintEntry *h = find_hint(which, P);
    if (!h || (!h->has_pos && !h->has_neg)) return false;
    const uint32_t nb = 1u << log_buckets(P);
    DepthHint d;
    d.npos = h->has_pos ? (h->has_neg ? nb >> 1 : nb) : 0u;
    d.nneg = nb - d.npos;
    d.plo = h->plo;
    d.nlo = h->nlo;
    d.pscale = (h->has_pos && h->phi > h->plo) ? (float)(d.npos - 1u) / (h->phi - h->plo) : 0.f;
    d.nscale = (h->has_neg && h->nhi > h->nlo) ? (float)(d.nneg - 1u) / (h->nhi - h->nlo) : 0.f;
    if (!std::isfinite(d.pscale) || !std::isfinite(d.nscale) || !std::isfinite(d.plo) || !std::isfinite(d.nlo)) return false;
    *out = d;
    return true;
}

void depth_hint_update(int which, size_t P, const uint32_t w[DW_COUNT], bool overflowed)
{
    if (which < 0 || which > 1) return;
    auto as_f = [](uint32_t bits) { union { uint32_t u; float f; } c; c.u = bits & 0x7FFFFFFFu; return c.f; };
    const bool has_pos = (w[DW_PMAX] | w[DW_PNMAX]) != 0u, has_neg = (w[DW_NMAX] | w[DW_NNMAX]) != 0u;
    // observed value ranges, padded by a quarter of their width (views of one scene see similar, not identical, ranges)
    float plo = as_f(~w[DW_PNMAX]), phi = as_f(w[DW_PMAX]), nlo = as_f(~w[DW_NNMAX]), nhi = as_f(w[DW_NMAX]);
    auto pad = [](float &lo, float &hi) {
        const float wdt = hi - lo, p = 0.25f * wdt + 1e-6f * fabsf(hi) + 1e-30f;
        lo -= p;
        hi += p;
    };
    if (has_pos) pad(plo, phi);
    if (has_neg) pad(nlo, nhi);
    HintEntry *h = find_hint(which, P);
    if (!h) {   // replace the least recently used entry
        h = &g_hints[which][0];
        for (auto &e : g_hints[which])
            if (!e.valid || e.age < h->age) { h = &e; if (!e.valid) break; }
        *h = HintEntry();
        h->P = P;
    }
    const bool fresh = !h->valid || overflowed || (g_hint_clock[which] & 63u) == 0u;
    auto merge = [&](bool has, bool &hhas, float &hlo, float &hhi, float lo, float hi) {
        if (!has) { if (fresh) hhas = false; return; }
        // widen an existing range (the union keeps several views' ranges covered) unless it has become much wider than
        // what is actually seen
        if (!fresh && hhas && (hhi - hlo) <= 8.0f * (hi - lo)) { hlo = fminf(hlo, lo); hhi = fmaxf(hhi, hi); }
        else { hlo = lo; hhi = hi; }
        hhas = true;
    };
    merge(has_pos, h->has_pos, h->plo, h->phi, plo, phi);
    merge(has_neg, h->has_neg, h->nlo, h->nhi, nlo, nhi);
    h->valid = true;
    h->age = ++g_hint_clock[which];
}

}  // namespace r2

// 0: never use depth hints, 1: use them (default; R2_DEPTH_HINT=0 in the environment also switches them off),
// 2: forget the hint history.  Results never depend on this: it only selects between two exact sorting paths.
extern "C" void r2_depth_hint_control(int mode)
{
    if (mode == 0 || mode == 1) r2::g_hint_mode.store(mode, std::memory_order_relaxed);
    if (mode == 2)
        for (auto &tab : r2::g_hints)
            for (auto &e : tab) e = r2::HintEntry();
}
