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

namespace r2 {

namespace {

constexpr uint32_t CULLED_KEY = 0xFFFFFFFFu;
constexpr uint32_t MAX_BUCKET = 256;  // a fuller bucket raises the fallback flag (ranking costs one pass over the bucket per key)

struct Ctrl {          // zeroed before every use
    uint32_t pmax;     // max / (complement of the) min of the keys with a clear sign bit (positive floats) ...
    uint32_t pnmax;
    uint32_t nmax;     // ... and of the keys with the sign bit set (negative floats: bits grow with |value|)
    uint32_t nnmax;
    uint32_t nculled;  // ticket counter of culled Gaussians
    uint32_t user;     // a word the caller's producer kernel may set (zeroed by depth_order_prepare); copied next to the
                       // overflow flag so that the host reads it back with the same 12-byte copy
    uint32_t pad[2];
};

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
__global__ void __launch_bounds__(1024) minmax_kernel(const uint32_t *__restrict__ keys, uint32_t n, Ctrl *__restrict__ c,
                                                      uint32_t *__restrict__ overflow)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        overflow[0] = 0u;        // only ever set by the rank kernel, three launches later
        overflow[1] = c->user;   // the producer kernel (preprocess) has finished: publish its flag
    }
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
    if (i >= n) return;
    const uint32_t k = keys[i];
    if (k == CULLED_KEY) {
        order[n - 1u - atomicAdd(&c->nculled, 1u)] = i;   // tail, any order: culled Gaussians emit nothing
        return;
    }
    const uint32_t b = bucket_of(k, c, log_nb);
    const uint32_t v = atomicSub(&counts[b], 1u);          // v in [1, count]: a unique slot inside the bucket
    const uint32_t pos = incl[b] - v;
    slot_key[pos] = k;
    slot_id[pos] = i;
}

// `weights` (optional): the rank kernel also accumulates sum(weights[id]) per 4096-position group of the final order
// into `partial` -- the first half of the prefix sum the caller runs over weights[order[j]] next (scan_apply_only).
__global__ void __launch_bounds__(256) bucket_rank_kernel(uint32_t nb, Ctrl *__restrict__ c, uint32_t *__restrict__ overflow, int log_nb,
                                                          const uint32_t *__restrict__ incl, const uint32_t *__restrict__ slot_key,
                                                          const uint32_t *__restrict__ slot_id, uint32_t *__restrict__ order,
                                                          const uint32_t *__restrict__ weights, uint32_t *__restrict__ partial)
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
            *overflow = 1u;
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
    if (weights) {
        // one atomic per wave when the wave's final positions share a group (nearly always: fp stays inside the bucket)
        const uint32_t wgt = p < nvis ? weights[id] : 0u;
        const uint32_t g = fp >> 12;
        const unsigned long long livem = __ballot(p < nvis);
        const uint32_t g0 = livem ? (uint32_t)__shfl(g, __ffsll((long long)livem) - 1) : 0u;   // group of the first live lane
        if (__all(p >= nvis || g == g0)) {
            uint32_t sum = wgt;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
            if ((threadIdx.x & 63) == 0 && sum) atomicAdd(&partial[g0], sum);
        } else if (p < nvis && wgt) {
            atomicAdd(&partial[g], wgt);
        }
    }
}

struct Temp {
    Ctrl *ctrl;
    uint32_t *counts;     // [nb]   (ctrl and counts are zeroed by one memset)
    uint32_t *partial;    // [P/4096 + 1] per-group weight sums of the final order (zeroed with the counts)
    uint32_t *incl;       // [nb]
    uint32_t *slot_key;   // [P]
    uint32_t *slot_id;    // [P]
    char *scan_temp;
    size_t scan_bytes, zero_bytes, bytes;
    static Temp carve(char *chunk, size_t P, size_t nb)
    {
        Temp t;
        Bump b(chunk);
        t.ctrl = b.take<Ctrl>(8);               // 128 bytes: keeps counts on the next 128-byte boundary
        t.counts = b.take<uint32_t>(nb);
        t.partial = b.take<uint32_t>(P / 4096 + 2);
        t.zero_bytes = b.off;
        t.incl = b.take<uint32_t>(nb);
        t.slot_key = b.take<uint32_t>(P);
        t.slot_id = b.take<uint32_t>(P);
        t.scan_bytes = scan_temp_bytes((int)nb);
        t.scan_temp = b.take<char>(t.scan_bytes);
        t.bytes = b.total();
        return t;
    }
};

inline int log_buckets(size_t P)
{
    int l = 10;
    while (l < 22 && ((size_t)1 << l) < P) ++l;   // ~1 bucket per key, 2^10 .. 2^22 buckets
    return l;
}

}  // namespace

size_t depth_order_temp_bytes(size_t P) { return Temp::carve(nullptr, P, (size_t)1 << log_buckets(P)).bytes; }

// order[P] = Gaussian ids sorted by (key, id); culled ids (key 0xFFFFFFFF) at the tail in arbitrary order.
// *overflow_flag (a device word) is set to 0, and to 1 when the result is INVALID (fall back to the radix sort); it must
// be read after the stream has caught up.
// zeroes the counters; may be called BEFORE the kernel that produces the keys, which can then set *depth_order_user_word
int depth_order_prepare(void *temp, size_t temp_bytes, size_t P, hipStream_t s)
{
    if (P == 0) return 0;
    const Temp t = Temp::carve(reinterpret_cast<char *>(temp), P, (size_t)1 << log_buckets(P));
    if (t.bytes > temp_bytes) {
        set_error("depth_order_prepare: temp storage too small (%zu < %zu)", temp_bytes, t.bytes);
        return R2_ERR_INVALID;
    }
    R2_HIP_TRY(hipMemsetAsync(temp, 0, t.zero_bytes, s));
    return 0;
}
uint32_t *depth_order_user_word(void *temp, size_t P)
{
    return &Temp::carve(reinterpret_cast<char *>(temp), P, (size_t)1 << log_buckets(P)).ctrl->user;
}

int depth_order_buckets(void *temp, size_t temp_bytes, const uint32_t *keys, uint32_t *order, size_t P,
                        uint32_t *overflow_flag, hipStream_t s, const uint32_t *weights, const uint32_t **partial_out,
                        bool prepared)
{
    if (partial_out) *partial_out = nullptr;
    if (P == 0) return 0;
    const int log_nb = log_buckets(P);
    const size_t nb = (size_t)1 << log_nb;
    const Temp t = Temp::carve(reinterpret_cast<char *>(temp), P, nb);
    if (t.bytes > temp_bytes) {
        set_error("depth_order_buckets: temp storage too small (%zu < %zu)", temp_bytes, t.bytes);
        return R2_ERR_INVALID;
    }
    const unsigned grid = (unsigned)((P + 255) / 256);
    if (!prepared) R2_HIP_TRY(hipMemsetAsync(temp, 0, t.zero_bytes, s));
    minmax_kernel<<<dim3(std::min((unsigned)((P + 4095) / 4096), 64u)), dim3(1024), 0, s>>>(keys, (uint32_t)P, t.ctrl, overflow_flag);
    bucket_count_kernel<<<dim3(grid), dim3(256), 0, s>>>(keys, (uint32_t)P, t.ctrl, log_nb, t.counts);
    const int rc = inclusive_scan_u32(t.scan_temp, t.scan_bytes, t.counts, t.incl, (int)nb, s);
    if (rc) return rc;
    bucket_place_kernel<<<dim3(grid), dim3(256), 0, s>>>(keys, (uint32_t)P, t.ctrl, log_nb, t.counts, t.incl, t.slot_key,
                                                         t.slot_id, order);
    bucket_rank_kernel<<<dim3(grid), dim3(256), 0, s>>>((uint32_t)nb, t.ctrl, overflow_flag, log_nb, t.incl, t.slot_key, t.slot_id,
                                                        order, weights, t.partial);
    if (partial_out && weights) *partial_out = t.partial;
    R2_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace r2
