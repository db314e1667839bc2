// radix_sort.hip -- stable LSD radix sort of (u32 key, u32 value) pairs on key bits [0, end_bit), written for
// the binning pipeline's problem sizes (1e5 .. 1e7 pairs) on MI355X.  It replaces the library sort the
// reference calls (cub::DeviceRadixSort::SortPairs, RAS/rasterizer_impl.cu:301-306): at these sizes a
// generic device sort is launch/latency-bound (measured ~25 us per digit pass whatever the size), while the
// data would stream through HBM in 2-5 us.
//
// Structure ("onesweep" with decoupled look-back, one kernel per digit pass):
//   * one histogram kernel reads the keys ONCE and counts every digit place (the digits of the original keys
//     do not change between passes);
//   * per pass, a workgroup takes a ticket (tile id in scheduling order), ranks its keys, publishes its
//     per-digit counts, resolves the exclusive prefix over earlier tiles by looking back at their published
//     counts, and scatters.  Thread d of the workgroup owns digit d for the whole exchange.
//   * ranking is wave-synchronous and needs no atomics: a wave walks its keys 64 at a time; lanes holding the
//     same digit find each other with one __ballot per digit bit (64-wide match), the lowest peer bumps the
//     wave's LDS counter.  Keys keep their relative order -> the sort is stable.
//   * tiles communicate through one dword per (tile, digit): 2 flag bits + a 30-bit count, written and polled
//     with relaxed AGENT-scope atomics (the per-XCD L2s are not coherent; see the MI355X guide, G16).
//     Tickets guarantee that a tile only ever waits on tiles that already started.
#include "r2_common.hpp"
#include <algorithm>

namespace r2 {

namespace {

constexpr int SORT_THREADS = 256;      // histogram kernel
constexpr int PASS_THREADS = 1024;     // pass kernel: 16 waves per tile hide the ranking's dependent chains
constexpr int PASS_WAVES = PASS_THREADS / 64;
constexpr int MAX_PASSES = 4;
constexpr int MAX_RADIX = 256;
constexpr uint32_t FLAG_AGG = 1u << 30;    // value = this tile's count of the digit
constexpr uint32_t FLAG_INC = 2u << 30;    // value = inclusive count over tiles 0..this
constexpr uint32_t VALUE_MASK = (1u << 30) - 1;
constexpr uint32_t SPIN_LIMIT = 1u << 22;  // bounded wait: a logic error must not hang the GPU
constexpr int LOOKBACK_WINDOW = 64;   // polls in flight per digit: a pass costs ~ (tiles / window) agent-scope round trips

struct PassPlan {
    int npass;
    int shift[MAX_PASSES];
    int bits[MAX_PASSES];
};

inline PassPlan make_plan(int end_bit)
{
    PassPlan p;
    if (end_bit < 1) end_bit = 1;
    if (end_bit > 32) end_bit = 32;
    p.npass = (end_bit + 7) / 8;
    const int base = end_bit / p.npass, extra = end_bit % p.npass;
    int s = 0;
    for (int i = 0; i < p.npass; ++i) {
        p.bits[i] = base + (i < extra ? 1 : 0);
        p.shift[i] = s;
        s += p.bits[i];
    }
    return p;
}

inline int items_per_thread(size_t n)
{
    // Few, fat tiles: the look-back chain grows with the tile count and costs ~2 us of agent-scope round
    // trip per window of LOOKBACK_WINDOW tiles; 16 waves per tile keep the CU busy meanwhile.
    if (n <= (size_t)128 * 1024) return 2;
    if (n <= (size_t)2 * 1024 * 1024) return 8;
    return 16;
}

struct SortTemp {
    uint32_t *keys_alt, *vals_alt;
    uint32_t *ghist;    // [MAX_PASSES][MAX_RADIX]
    uint32_t *ticket;   // [MAX_PASSES]
    uint32_t *error;    // [1]
    uint32_t *status;   // [npass][ntiles][MAX_RADIX]
    size_t zero_off, zero_bytes, bytes;
    static SortTemp carve(char *chunk, size_t n, size_t ntiles)
    {
        SortTemp t;
        Bump b(chunk);
        t.keys_alt = b.take<uint32_t>(n);
        t.vals_alt = b.take<uint32_t>(n);
        t.zero_off = b.offset_of_next();
        t.ghist = b.take<uint32_t>(MAX_PASSES * MAX_RADIX);
        t.ticket = b.take<uint32_t>(MAX_PASSES);
        t.error = b.take<uint32_t>(1);
        t.status = b.take<uint32_t>((size_t)MAX_PASSES * ntiles * MAX_RADIX);
        t.zero_bytes = b.off - t.zero_off;
        t.bytes = b.total();
        return t;
    }
};

__device__ __forceinline__ uint32_t ld_agent(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint32_t *p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// counts every digit place of every key in one read of the key array.  The LDS histogram is replicated
// HIST_COPIES times (copy = lane & 7): depth keys share their high byte, and 64 lanes hammering one LDS
// counter serialise 64-way.
constexpr int HIST_COPIES = 8;
__global__ void __launch_bounds__(SORT_THREADS) radix_hist_kernel(const uint32_t *__restrict__ keys, uint32_t n,
                                                                  PassPlan plan, uint32_t *__restrict__ ghist)
{
    __shared__ uint32_t h[HIST_COPIES][MAX_PASSES * MAX_RADIX];
    for (int i = threadIdx.x; i < HIST_COPIES * MAX_PASSES * MAX_RADIX; i += SORT_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    const uint32_t stride = gridDim.x * SORT_THREADS;
    uint32_t *mine = h[threadIdx.x & (HIST_COPIES - 1)];
    for (uint32_t i = blockIdx.x * SORT_THREADS + threadIdx.x; i < n; i += stride) {
        const uint32_t k = keys[i];
#pragma unroll
        for (int p = 0; p < MAX_PASSES; ++p)
            if (p < plan.npass) atomicAdd(&mine[p * MAX_RADIX + ((k >> plan.shift[p]) & ((1u << plan.bits[p]) - 1u))], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MAX_PASSES * MAX_RADIX; i += SORT_THREADS) {
        uint32_t c = 0;
#pragma unroll
        for (int j = 0; j < HIST_COPIES; ++j) c += h[j][i];
        if (c) atomicAdd(&ghist[i], c);
    }
}

template <int IPT>
__global__ void __launch_bounds__(PASS_THREADS) radix_pass_kernel(
    const uint32_t *__restrict__ kin, const uint32_t *__restrict__ vin, uint32_t *__restrict__ kout,
    uint32_t *__restrict__ vout, uint32_t n, int shift, int bits, const uint32_t *__restrict__ ghist,
    uint32_t *__restrict__ ticket, uint32_t *__restrict__ status, uint32_t *__restrict__ error)
{
    constexpr uint32_t TILE = PASS_THREADS * IPT;
    __shared__ uint32_t wave_hist[PASS_WAVES][MAX_RADIX];   // per-wave digit counts, then exclusive wave offsets
    __shared__ uint32_t bin_base[MAX_RADIX];                // global position of the tile's first key per digit
    __shared__ uint32_t tile_start[MAX_RADIX];              // position of the digit's first key inside the tile
    __shared__ uint32_t scan_tmp[4], scan_tmp2[4];
    __shared__ uint32_t s_key[TILE], s_val[TILE];           // the tile, regrouped by digit, for coalesced stores
    __shared__ uint32_t s_tile;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t radix = 1u << bits;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    for (int i = tid; i < PASS_WAVES * MAX_RADIX; i += PASS_THREADS) (&wave_hist[0][0])[i] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t base = tile * TILE + (uint32_t)wave * (64u * IPT);

    // ---- load the tile's keys and values up front: every load is in flight before the first use
    uint32_t key[IPT], val[IPT], rank[IPT];
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const uint32_t idx = base + (uint32_t)i * 64u + (uint32_t)lane;
        key[i] = idx < n ? kin[idx] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const uint32_t idx = base + (uint32_t)i * 64u + (uint32_t)lane;
        val[i] = idx < n ? vin[idx] : 0u;
    }
    // ---- rank: wave-synchronous 64-wide digit matching, rows in key order.  The 8 ballots of a row are
    // independent (digit bits above `bits` are 0 in every lane, so their term is all-ones) and are combined
    // with a tree of ANDs: the dependent chain per row is ~6 instructions, 16 waves per CU cover its latency.
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const uint32_t idx = base + (uint32_t)i * 64u + (uint32_t)lane;
        const bool valid = idx < n;
        const uint32_t d = (key[i] >> shift) & (radix - 1u);
        unsigned long long m[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long vote = __ballot(bit);
            m[b] = bit ? vote : ~vote;
        }
        const unsigned long long peers =
            __ballot(valid) & ((m[0] & m[1]) & (m[2] & m[3])) & ((m[4] & m[5]) & (m[6] & m[7]));
        const uint32_t before = wave_hist[wave][d];
        rank[i] = before + (uint32_t)__popcll(peers & lt_mask);
        if (valid && (peers & lt_mask) == 0ull) wave_hist[wave][d] = before + (uint32_t)__popcll(peers);
    }
    __syncthreads();

    // ---- thread d owns digit d: wave offsets, tile total, look-back, global base
    uint32_t total = 0, gcount = 0;
    if ((uint32_t)tid < radix) {
#pragma unroll
        for (int w = 0; w < PASS_WAVES; ++w) {
            const uint32_t c = wave_hist[w][tid];
            wave_hist[w][tid] = total;
            total += c;
        }
        gcount = ghist[tid];
        st_agent(&status[(size_t)tile * MAX_RADIX + tid], (tile == 0 ? FLAG_INC : FLAG_AGG) | total);
    }
    // exclusive scan of the global digit histogram (first 256 threads) -> first output position of each digit
    // ... and of the tile's own digit counts -> where each digit's run starts inside the regrouped tile
    uint32_t digit_start = 0, local_start = 0;
    if (tid < MAX_RADIX) {
        uint32_t incl = gcount, incl2 = total;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d), up2 = __shfl_up(incl2, d);
            if (lane >= d) { incl += up; incl2 += up2; }
        }
        if (lane == 63) { scan_tmp[wave] = incl; scan_tmp2[wave] = incl2; }
        digit_start = incl - gcount;
        local_start = incl2 - total;
    }
    __syncthreads();
    if (tid < MAX_RADIX) {
        for (int w = 0; w < wave; ++w) { digit_start += scan_tmp[w]; local_start += scan_tmp2[w]; }
        tile_start[tid] = local_start;
    }
    __syncthreads();
    // regroup the tile by digit in LDS while the look-back is in flight (stable: rank keeps key order)
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const uint32_t idx = base + (uint32_t)i * 64u + (uint32_t)lane;
        if (idx < n) {
            const uint32_t d = (key[i] >> shift) & (radix - 1u);
            const uint32_t pos = tile_start[d] + wave_hist[wave][d] + rank[i];
            s_key[pos] = key[i];
            s_val[pos] = val[i];
        }
    }

    if ((uint32_t)tid < radix) {
        uint32_t excl = 0;
        if (tile > 0) {
            // walk back over earlier tiles, LOOKBACK_WINDOW independent polls in flight at a time: a tile that
            // already resolved its own prefix (FLAG_INC) ends the walk, aggregates are summed on the way
            uint32_t t = tile, spins = 0;
            bool done = false;
            while (!done) {
                const uint32_t wlen = min(t, (uint32_t)LOOKBACK_WINDOW);
                uint32_t sv[LOOKBACK_WINDOW];
#pragma unroll
                for (int j = 0; j < LOOKBACK_WINDOW; ++j)
                    sv[j] = (uint32_t)j < wlen ? ld_agent(&status[(size_t)(t - 1 - j) * MAX_RADIX + tid]) : FLAG_INC;
                uint32_t used = 0;
#pragma unroll
                for (int j = 0; j < LOOKBACK_WINDOW; ++j) {
                    if (done || used != (uint32_t)j) continue;   // stop at the first unpublished entry
                    if (sv[j] == 0) continue;
                    excl += sv[j] & VALUE_MASK;
                    ++used;
                    if (sv[j] & FLAG_INC) done = true;
                }
                t -= min(used, wlen);
                if (t == 0) done = true;
                if (!done && used == 0) {
                    if (++spins > SPIN_LIMIT) { atomicOr(error, 1u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            st_agent(&status[(size_t)tile * MAX_RADIX + tid], FLAG_INC | (excl + total));
        }
        bin_base[tid] = digit_start + excl;
    }
    __syncthreads();

    // ---- store: slot j of the regrouped tile goes to bin_base[d] + (j - tile_start[d]); consecutive slots of
    // one digit are consecutive in memory, so a wave writes a few contiguous runs instead of 64 scattered words
    const uint32_t tile_n = min(TILE, n - tile * TILE);
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const uint32_t j = (uint32_t)i * PASS_THREADS + (uint32_t)tid;
        if (j < tile_n) {
            const uint32_t k = s_key[j];
            const uint32_t d = (k >> shift) & (radix - 1u);
            const uint32_t dst = bin_base[d] + (j - tile_start[d]);
            kout[dst] = k;
            vout[dst] = s_val[j];
        }
    }
}

template <int IPT>
void launch_pass(const uint32_t *kin, const uint32_t *vin, uint32_t *kout, uint32_t *vout, uint32_t n, int shift,
                 int bits, const uint32_t *ghist, uint32_t *ticket, uint32_t *status, uint32_t *error, uint32_t ntiles,
                 hipStream_t s)
{
    radix_pass_kernel<IPT><<<dim3(ntiles), dim3(PASS_THREADS), 0, s>>>(kin, vin, kout, vout, n, shift, bits, ghist,
                                                                       ticket, status, error);
}

}  // namespace

size_t sort_temp_bytes(size_t n)
{
    const size_t tile = (size_t)PASS_THREADS * items_per_thread(n);
    return SortTemp::carve(nullptr, n, (n + tile - 1) / tile).bytes;
}

int sort_pairs_u32_u32(void *temp, size_t temp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin,
                       uint32_t *vout, size_t n, int end_bit, hipStream_t s)
{
    if (n == 0) return 0;
    if (n >= (size_t)VALUE_MASK) {
        set_error("sort_pairs_u32_u32: %zu pairs exceed the 2^30 look-back counter range", n);
        return R2_ERR_INVALID;
    }
    const int ipt = items_per_thread(n);
    const size_t tile = (size_t)PASS_THREADS * ipt;
    const uint32_t ntiles = (uint32_t)((n + tile - 1) / tile);
    const SortTemp t = SortTemp::carve(reinterpret_cast<char *>(temp), n, ntiles);
    if (t.bytes > temp_bytes) {
        set_error("sort_pairs_u32_u32: temp storage too small (%zu < %zu)", temp_bytes, t.bytes);
        return R2_ERR_INVALID;
    }
    const PassPlan plan = make_plan(end_bit);
    R2_HIP_TRY(hipMemsetAsync(reinterpret_cast<char *>(temp) + t.zero_off, 0, t.zero_bytes, s));
    const uint32_t hist_blocks = (uint32_t)std::min<size_t>(512, (n + SORT_THREADS * 4 - 1) / (SORT_THREADS * 4));
    radix_hist_kernel<<<dim3(hist_blocks), dim3(SORT_THREADS), 0, s>>>(kin, (uint32_t)n, plan, t.ghist);
    const uint32_t *src_k = kin, *src_v = vin;
    for (int p = 0; p < plan.npass; ++p) {
        const bool to_out = ((plan.npass - p) & 1) != 0;   // the last pass lands in (kout, vout)
        uint32_t *dst_k = to_out ? kout : t.keys_alt, *dst_v = to_out ? vout : t.vals_alt;
        uint32_t *st = t.status + (size_t)p * ntiles * MAX_RADIX;
        const uint32_t *gh = t.ghist + p * MAX_RADIX;
        switch (ipt) {
        case 2: launch_pass<2>(src_k, src_v, dst_k, dst_v, (uint32_t)n, plan.shift[p], plan.bits[p], gh, t.ticket + p, st, t.error, ntiles, s); break;
        case 8: launch_pass<8>(src_k, src_v, dst_k, dst_v, (uint32_t)n, plan.shift[p], plan.bits[p], gh, t.ticket + p, st, t.error, ntiles, s); break;
        default: launch_pass<16>(src_k, src_v, dst_k, dst_v, (uint32_t)n, plan.shift[p], plan.bits[p], gh, t.ticket + p, st, t.error, ntiles, s); break;
        }
        src_k = dst_k;
        src_v = dst_v;
    }
    R2_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace r2
