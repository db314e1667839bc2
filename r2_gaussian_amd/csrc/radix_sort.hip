// radix_sort.hip -- stable LSD radix sort of (u32 key, u32 value[, u32 value2]) on key bits [0, end_bit), written for
// the binning pipeline's problem sizes (1e5 .. 1e7 pairs) on MI355X.  It replaces the library sort the reference calls
// (cub::DeviceRadixSort::SortPairs, RAS/rasterizer_impl.cu:301-306).
//
// At these sizes a sort is bound by LAUNCH COUNT and dependent memory round trips, not by bandwidth (the data streams
// through HBM in a few microseconds), so the design minimises digit passes:
//   * digits are up to 12 bits wide (4096 bins): the 11-bit tile ids of a 512^2 detector sort in ONE pass, 32-bit depth
//     keys in three (12 + 12 + 8);
//   * a pass whose digit is the same in every key is detected on the device and skipped (optional): the exponent byte of
//     cone-beam depths is constant, which leaves two passes for the depth order;
//   * a pass is three kernels without any inter-workgroup waiting (no look-back spinning, nothing to deadlock):
//       upsweep   one workgroup per 4096-key tile counts its digits (LDS atomics) -> H[tile][digit]
//       scan      column-wise exclusive prefix of H over the tiles (32 digits x 8 tile-slices per workgroup) + totals
//       downsweep re-reads the tile, ranks it (wave-synchronous 64-wide digit matching with __ballot: no atomics,
//                 keys keep their order -> stable) and scatters to  digit_base + H[tile][digit] + rank
//   * for a single-pass sort the digit totals ARE the bucket sizes: the caller gets them back and derives the tile
//     ranges from them instead of scanning the sorted keys for boundaries (identifyTileRanges, RAS/rasterizer_impl.cu:116-138).
#include "r2_common.hpp"
#include <algorithm>

R2_TS_DEFINE(sort)

namespace r2 {

// Experiment builds (python -m r2_gaussian_amd.build -DR2_EXP_TS --out=libr2hip_ts.so; scripts/cbench prints the table):
// s_memrealtime stamps per phase and workgroup -- the timeline INSIDE and BETWEEN the three kernels of a pass.  What it showed
// for the single-pass tile sort of 1.16 M instances (284 workgroups, 33 us from the upsweep's launch to the downsweep's end):
// upsweep 0 -> 3.3 us, 1 us gap, scan 4.5 -> 8.1, 1.8 us gap, downsweep 10.0 -> 21.5 (median workgroup: loads 3.8, ranking 3.8,
// per-digit epilogue 1.7, scatter 1.5) but -> 27.5 for the 28 CUs that host two workgroups, which is the kernel's end.
// Tried on that evidence, none better than +-1 us: 2048-key tiles (twice the workgroups, half the chain), 1024-thread
// workgroups, peer masks of all rows before the LDS chain, all loads issued up front, the range/work-list job in an extra
// workgroup instead of workgroup 0's prologue (7.4 us, but the two-workgroup CUs finish later anyway).
#define R2_TS(ph) R2_TS_AT(sort, ph)

namespace {

constexpr int RS_THREADS = 512;                // 8 waves rank a tile side by side (the per-wave ranking chain is serial)
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_IPT = 8;
constexpr int RS_TILE = RS_THREADS * RS_IPT;   // 4096 keys per workgroup
constexpr int RS_SCAN_THREADS = 1024;             // 32 digits x 32 tile slices
constexpr int RS_MAX_BITS = 12;
constexpr int RS_MAX_RADIX = 1 << RS_MAX_BITS;
constexpr int RS_MAX_PASSES = 4;
constexpr int RS_SCAN_DIGITS = 32;             // digits per scan workgroup (one 128-byte row segment of H)
constexpr int RS_SCAN_ROWS = RS_SCAN_THREADS / RS_SCAN_DIGITS;
#ifndef R2_RS_SCAN_BATCH
#define R2_RS_SCAN_BATCH 16
#endif
constexpr int RS_SCAN_BATCH = R2_RS_SCAN_BATCH;   // tile rows a scan thread requests at once

struct Plan {
    int npass;
    int shift[RS_MAX_PASSES];
    int bits[RS_MAX_PASSES];
};

inline Plan make_plan(int end_bit)
{
    Plan p;
    end_bit = std::min(32, std::max(1, end_bit));
    if (end_bit == 32) {   // 12 + 12 + 8: the top pass covers exactly the float exponent byte (skippable, see header)
        p.npass = 3;
        p.shift[0] = 0; p.bits[0] = 12;
        p.shift[1] = 12; p.bits[1] = 12;
        p.shift[2] = 24; p.bits[2] = 8;
        return p;
    }
    p.npass = (end_bit + RS_MAX_BITS - 1) / RS_MAX_BITS;
    const int base = end_bit / p.npass, extra = end_bit % p.npass;
    int s = 0;
    for (int i = 0; i < p.npass; ++i) {
        p.bits[i] = base + (i < extra ? 1 : 0);
        p.shift[i] = s;
        s += p.bits[i];
    }
    return p;
}

// the three buffer sets a sort moves between: 0 = caller's input (read-only), 1 = scratch, 2 = caller's output
struct Buffers {
    const uint32_t *k0, *v0, *w0;
    uint32_t *k1, *v1, *w1;
    uint32_t *k2, *v2, *w2;
};
__device__ __forceinline__ const uint32_t *rd_k(const Buffers &b, int i) { return i == 0 ? b.k0 : (i == 1 ? b.k1 : b.k2); }
__device__ __forceinline__ const uint32_t *rd_v(const Buffers &b, int i) { return i == 0 ? b.v0 : (i == 1 ? b.v1 : b.v2); }
__device__ __forceinline__ const uint32_t *rd_w(const Buffers &b, int i) { return i == 0 ? b.w0 : (i == 1 ? b.w1 : b.w2); }
__device__ __forceinline__ uint32_t *wr_k(const Buffers &b, int i) { return i == 1 ? b.k1 : b.k2; }
__device__ __forceinline__ uint32_t *wr_v(const Buffers &b, int i) { return i == 1 ? b.v1 : b.v2; }
__device__ __forceinline__ uint32_t *wr_w(const Buffers &b, int i) { return i == 1 ? b.w1 : b.w2; }

// The e-th EXECUTED pass writes buffer set target(e) and reads target(e-1) (the input for e == 0).  `phase` is chosen
// on the host so that a sort that executes every pass ends in the output set.
__device__ __host__ __forceinline__ int target_of(int e, int phase) { return ((e + phase) & 1) ? 2 : 1; }
__device__ __forceinline__ int executed_before(const uint32_t *skip, int pass)
{
    int e = 0;
    for (int q = 0; q < pass; ++q) e += skip[q] ? 0 : 1;
    return e;
}

// LDS digit arrays are padded by one word per 32 digits: the per-digit epilogue walks them with a stride of radix/256
// words per lane, which would otherwise land on 2 of the 32 banks.
__device__ __host__ __forceinline__ uint32_t pidx(uint32_t d) { return d + (d >> 5); }

// ---------------------------------------------------------------------------------------------------------- upsweep
__global__ void __launch_bounds__(RS_THREADS) rs_upsweep_kernel(Buffers buf, uint32_t n, int shift, int bits, int pass,
                                                                int phase, const uint32_t *__restrict__ skip,
                                                                uint32_t *__restrict__ H, uint32_t *__restrict__ clear_skip,
                                                                int ipt = RS_IPT /* rows of RS_THREADS keys per workgroup */)
{
    __shared__ uint32_t hist[RS_MAX_RADIX];
    R2_TS(8);
    const uint32_t radix = 1u << bits;
    const int e = executed_before(skip, pass);
    const uint32_t *__restrict__ keys = rd_k(buf, e == 0 ? 0 : target_of(e - 1, phase));
    for (uint32_t d = threadIdx.x; d < radix; d += RS_THREADS) hist[d] = 0;
    // (first pass of a sort: the skip words start out clear -- spares a memset launch; the word of this pass is set by the scan
    // kernel that follows, the others by later passes)
    if (clear_skip && blockIdx.x == 0 && threadIdx.x < RS_MAX_PASSES * 2) clear_skip[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * (uint32_t)(ipt * RS_THREADS);
    if (ipt == RS_IPT) {
#pragma unroll
        for (int i = 0; i < RS_IPT; ++i) {
            const uint32_t idx = base + (uint32_t)i * RS_THREADS + threadIdx.x;
            if (idx < n) atomicAdd(&hist[(keys[idx] >> shift) & (radix - 1u)], 1u);
        }
    } else {
        for (int i = 0; i < ipt; ++i) {
            const uint32_t idx = base + (uint32_t)i * RS_THREADS + threadIdx.x;
            if (idx < n) atomicAdd(&hist[(keys[idx] >> shift) & (radix - 1u)], 1u);
        }
    }
    __syncthreads();
    uint32_t *__restrict__ row = H + (size_t)blockIdx.x * radix;
    for (uint32_t d = threadIdx.x; d < radix; d += RS_THREADS) row[d] = hist[d];
    R2_TS(9);
}

// ------------------------------------------------------------------------------------------------------------- scan
// (Measured and left out, round 3: 8 digits per workgroup for narrow radices -- 32 workgroups instead of 8 for the voxelizer's
// 8-bit passes, a quarter of the dependent row loads per thread: 12 us SLOWER per pass; a wave then touches eight 32-byte
// sectors per load instead of two 128-byte lines.)
// H[t][d] <- sum of H[t'][d] over t' < t (exclusive, per digit); totals[d] = column sum.  One workgroup owns 32
// consecutive digits; its 8 thread rows split the tile range, so every access is a full 128-byte row segment.
__global__ void __launch_bounds__(RS_SCAN_THREADS) rs_scan_kernel(uint32_t *__restrict__ H, uint32_t ntiles, int bits, uint32_t n,
                                                             int pass, int allow_skip, uint32_t *__restrict__ skip,
                                                             uint32_t *__restrict__ totals)
{
    __shared__ uint32_t part[RS_SCAN_ROWS][RS_SCAN_DIGITS];
    R2_TS(10);
    const uint32_t radix = 1u << bits;
    const uint32_t dl = threadIdx.x % RS_SCAN_DIGITS, row = threadIdx.x / RS_SCAN_DIGITS;
    const uint32_t d = blockIdx.x * RS_SCAN_DIGITS + dl;
    const uint32_t per = (ntiles + RS_SCAN_ROWS - 1) / RS_SCAN_ROWS;
    const uint32_t t0 = row * per, t1 = min(ntiles, t0 + per);
    uint32_t sum = 0;
    if (d < radix) {   // RS_SCAN_BATCH loads in flight at a time (one load per iteration behind a counted wait was half of this kernel)
        for (uint32_t t = t0; t < t1; t += RS_SCAN_BATCH) {
            uint32_t v[RS_SCAN_BATCH];
#pragma unroll
            for (int u = 0; u < RS_SCAN_BATCH; ++u) v[u] = H[(size_t)min(t + (uint32_t)u, t1 - 1u) * radix + d];   // branch-free: clamped
#pragma unroll
            for (int u = 0; u < RS_SCAN_BATCH; ++u) sum += (t + (uint32_t)u < t1) ? v[u] : 0u;
        }
    }
    part[row][dl] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (int r = 0; r < RS_SCAN_ROWS; ++r) {
        const uint32_t v = part[r][dl];
        if ((uint32_t)r < row) run += v;
        total += v;
    }
    if (d < radix) {
        for (uint32_t t = t0; t < t1; t += RS_SCAN_BATCH) {   // RS_SCAN_BATCH loads in flight, then the dependent stores
            uint32_t v[RS_SCAN_BATCH];
#pragma unroll
            for (int u = 0; u < RS_SCAN_BATCH; ++u) v[u] = H[(size_t)min(t + (uint32_t)u, t1 - 1u) * radix + d];   // branch-free: clamped
#pragma unroll
            for (int u = 0; u < RS_SCAN_BATCH; ++u) {
                if (t + u < t1) {
                    H[(size_t)(t + u) * radix + d] = run;
                    run += v[u];
                }
            }
        }
        if (row == 0) {
            totals[d] = total;
            if (allow_skip && total == n) skip[pass] = 1u;   // every key has this digit: the pass would be the identity
        }
    }
    R2_TS(11);
}

// -------------------------------------------------------------------------------------------------------- downsweep
// INV (single-pass sorts only): instead of scattering keys and the identity payload (out[pos] = index), write the INVERSE
// permutation inv[index] = pos -- a coalesced store -- and scatter only the second payload.  Scattered 4-byte stores
// from 8 XCDs into the same cache lines are the expensive part of a wide-digit pass; this cuts them by 3x.
// INVV (last pass of a multi-pass sort whose value payload is the original index): write inv[value] = pos instead of
// out[pos] = value -- the same number of scattered stores, but it replaces a separate permutation-inversion kernel.
// IPT: rows of 64 keys per wave, i.e. a workgroup sorts IPT * RS_THREADS consecutive keys.  The single-pass tile sort picks it so
// that the workgroups fill the CUs in whole rounds (284 workgroups of 4096 keys on 256 CUs ran as long as 512 would: the CUs
// hosting two of them finish last; 252 workgroups of 4608 keys are one round).
template <bool HAS_W, bool INV, bool INVV = false, int IPT = RS_IPT>
__global__ void __launch_bounds__(RS_THREADS) rs_downsweep_kernel(Buffers buf, uint32_t n, int shift, int bits, int pass,
                                                                  int phase, const uint32_t *__restrict__ skip,
                                                                  const uint32_t *__restrict__ H,
                                                                  const uint32_t *__restrict__ totals,
                                                                  uint32_t *__restrict__ inv, WorkListOut wo)
{
    extern __shared__ uint32_t smem[];
    const uint32_t radix = 1u << bits;
    const uint32_t prad = pidx(radix);                 // padded row length
    uint32_t *wave_hist = smem;                        // [RS_WAVES][prad] per-wave digit counts -> exclusive wave offsets
    uint32_t *digit_off = smem + RS_WAVES * prad;      // [prad] global position of this tile's first key of each digit
    __shared__ uint32_t wsum[RS_WAVES];

    constexpr uint32_t TILE = (uint32_t)IPT * RS_THREADS;   // keys per workgroup
    R2_TS(0);
    if (skip[pass]) return;
    // single-pass tile sort: the digit totals are the per-tile instance counts -- ONE EXTRA workgroup (the last) turns them
    // into the tile ranges and the render kernels' work list (one launch less on the forward's critical path; as a
    // prologue of workgroup 0 the 7 us job delayed that workgroup's own tile, which is the kernel's end once the
    // workgroups run in a single round)
    if (INV && wo.ranges != nullptr && blockIdx.x == gridDim.x - 1) {
        ranges_and_work_block<RS_THREADS>(totals, wo);
        return;
    }
    R2_TS(1);
    // Buffer set of this pass.  Selecting among the three sets with a run-time index made the compiler copy the nine
    // pointers to scratch (80 bytes per lane: a kernel with scratch pays ~4 us more at dispatch); the single-pass sort
    // knows its sets at compile time, the general passes pick their pointers with scalar selects.
    const int e = INV ? 0 : executed_before(skip, pass);
    const int src = e == 0 ? 0 : target_of(e - 1, phase), dst = INV ? 2 : target_of(e, phase);
    const bool s0 = src == 0, s1 = src == 1, d1 = dst == 1;
    const uint32_t *__restrict__ kin = INV ? buf.k0 : (s0 ? buf.k0 : (s1 ? (const uint32_t *)buf.k1 : (const uint32_t *)buf.k2));
    const uint32_t *__restrict__ vin = INV ? buf.v0 : (s0 ? buf.v0 : (s1 ? (const uint32_t *)buf.v1 : (const uint32_t *)buf.v2));
    const uint32_t *__restrict__ win = INV ? buf.w0 : (s0 ? buf.w0 : (s1 ? (const uint32_t *)buf.w1 : (const uint32_t *)buf.w2));
    uint32_t *__restrict__ kout = INV ? buf.k2 : (d1 ? buf.k1 : buf.k2);
    uint32_t *__restrict__ vout = INV ? buf.v2 : (d1 ? buf.v1 : buf.v2);
    uint32_t *__restrict__ wout = INV ? buf.w2 : (d1 ? buf.w1 : buf.w2);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t i = tid; i < RS_WAVES * prad; i += RS_THREADS) wave_hist[i] = 0;
    const uint32_t tile = blockIdx.x;
    const uint32_t base = tile * TILE + (uint32_t)wave * (64u * IPT);

    // ---- load the wave's 64 * IPT consecutive keys (rows of 64), every load in flight before the first use
    uint32_t key[IPT], val[IPT], wal[IPT], rank[IPT];
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const uint32_t idx = base + (uint32_t)i * 64u + (uint32_t)lane;
        key[i] = idx < n ? kin[idx] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const uint32_t idx = base + (uint32_t)i * 64u + (uint32_t)lane;
        val[i] = (!INV && vin && idx < n) ? vin[idx] : idx;   // absent value array = identity (only ever the input set)
        wal[i] = (HAS_W && idx < n) ? win[idx] : 0u;
    }
    __syncthreads();
    R2_TS(2);
    // ---- rank: rows in key order; lanes holding the same digit find each other with one ballot per digit bit
    uint32_t *wh = wave_hist + wave * prad;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const uint32_t idx = base + (uint32_t)i * 64u + (uint32_t)lane;
        const bool valid = idx < n;
        const uint32_t d = (key[i] >> shift) & (radix - 1u);
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < RS_MAX_BITS; ++b) {
            if (b < bits) {   // wave-uniform
                const bool bit = (d >> b) & 1u;
                const unsigned long long vote = __ballot(bit);
                peers &= bit ? vote : ~vote;
            }
        }
        const uint32_t before = wh[pidx(d)];
        rank[i] = before + (uint32_t)__popcll(peers & lt_mask);
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & lt_mask) == 0ull) wh[pidx(d)] = before + (uint32_t)__popcll(peers);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    R2_TS(3);

    // ---- per digit: exclusive offsets of the waves inside the tile; digit base = exclusive scan of the totals
    const uint32_t dpt = radix / RS_THREADS > 0 ? radix / RS_THREADS : 1;   // consecutive digits per thread
    uint32_t tsum = 0;
    for (uint32_t j = 0; j < dpt; ++j) {
        const uint32_t d = tid * dpt + j;
        if (d < radix) tsum += totals[d];
    }
    uint32_t incl = tsum;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const uint32_t up = __shfl_up(incl, s);
        if (lane >= s) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t dbase = incl - tsum;
    for (int w = 0; w < wave; ++w) dbase += wsum[w];
    const uint32_t *__restrict__ hrow = H + (size_t)tile * radix;
    uint32_t mytot = 0;   // keys of this tile holding this thread's digit(s)
    for (uint32_t j = 0; j < dpt; ++j) {
        const uint32_t d = tid * dpt + j;
        if (d < radix) {
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < RS_WAVES; ++w) {
                const uint32_t c = wave_hist[w * prad + pidx(d)];
                wave_hist[w * prad + pidx(d)] = run;
                run += c;
            }
            digit_off[pidx(d)] = dbase + hrow[d];
            dbase += totals[d];
            mytot += run;
        }
    }
    __syncthreads();
    R2_TS(4);

    // Narrow digits (<= 8 bits: a 4096-key tile holds ~16 keys per digit): regroup the tile by digit in LDS first, so
    // that a wave's stores fall into a few contiguous runs instead of 64 scattered words -- scattered 4-byte stores from
    // 8 XCDs into shared cache lines are what a radix pass costs on this machine.
    const bool regroup = !INV && bits <= 8;   // wave-uniform, radix <= 256 <= RS_THREADS: one digit per thread
    if (regroup) {
        uint32_t *tstart = digit_off + prad;              // [prad] first slot of each digit inside the regrouped tile
        uint32_t *sK = tstart + prad, *sV = sK + TILE, *sW = sV + TILE;
        uint32_t incl2 = mytot;
#pragma unroll
        for (int s2 = 1; s2 < 64; s2 <<= 1) {
            const uint32_t up = __shfl_up(incl2, s2);
            if (lane >= s2) incl2 += up;
        }
        if (lane == 63) wsum[wave] = incl2;
        __syncthreads();
        uint32_t lstart = incl2 - mytot;
        for (int w = 0; w < wave; ++w) lstart += wsum[w];
        if ((uint32_t)tid < radix) tstart[pidx(tid)] = lstart;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const uint32_t idx = base + (uint32_t)i * 64u + (uint32_t)lane;
            if (idx < n) {
                const uint32_t d = (key[i] >> shift) & (radix - 1u);
                const uint32_t li = tstart[pidx(d)] + wh[pidx(d)] + rank[i];
                sK[li] = key[i];
                sV[li] = val[i];
                if (HAS_W) sW[li] = wal[i];
            }
        }
        __syncthreads();
        const uint32_t tile_n = min(TILE, n - tile * TILE);
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const uint32_t j = (uint32_t)i * RS_THREADS + (uint32_t)tid;
            if (j < tile_n) {
                const uint32_t k = sK[j];
                const uint32_t d = (k >> shift) & (radix - 1u);
                const uint32_t pos = digit_off[pidx(d)] + (j - tstart[pidx(d)]);
                kout[pos] = k;
                if (INVV) inv[sV[j]] = pos;
                else vout[pos] = sV[j];
                if (HAS_W) wout[pos] = sW[j];
            }
        }
        return;
    }

    // ---- scatter
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const uint32_t idx = base + (uint32_t)i * 64u + (uint32_t)lane;
        if (idx < n) {
            const uint32_t d = (key[i] >> shift) & (radix - 1u);
            const uint32_t pos = digit_off[pidx(d)] + wh[pidx(d)] + rank[i];
            if (INV) {
                if (inv != nullptr) inv[idx] = pos;   // introspection only (debug mode)
            } else {
                kout[pos] = key[i];
                if (INVV) inv[val[i]] = pos;
                else vout[pos] = val[i];
            }
            if (HAS_W) wout[pos] = wal[i];
        }
    }
    R2_TS(5);
}

// ---------------------------------------------------------------------------------------------------------- finalize
// With pass skipping the result may sit in the input or the scratch set: move it to the output set.
template <bool HAS_W>
__global__ void __launch_bounds__(RS_THREADS) rs_finalize_kernel(Buffers buf, uint32_t n, int npass, int phase,
                                                                 const uint32_t *__restrict__ skip)
{
    const int E = executed_before(skip, npass);
    const int loc = E == 0 ? 0 : target_of(E - 1, phase);
    if (loc == 2) return;
    const uint32_t *__restrict__ k = rd_k(buf, loc);
    const uint32_t *__restrict__ v = rd_v(buf, loc);
    const uint32_t *__restrict__ w = rd_w(buf, loc);
    for (uint32_t i = blockIdx.x * RS_THREADS + threadIdx.x; i < n; i += gridDim.x * RS_THREADS) {
        buf.k2[i] = k[i];
        buf.v2[i] = v ? v[i] : i;
        if (HAS_W) buf.w2[i] = w[i];
    }
}

struct SortTemp {
    uint32_t *keys_alt, *vals_alt, *vals2_alt;
    uint32_t *H;        // [ntiles][radix_max_used]
    uint32_t *totals;   // [RS_MAX_RADIX]
    uint32_t *skip;     // [RS_MAX_PASSES]
    size_t bytes;
    static SortTemp carve(char *chunk, size_t n)
    {
        SortTemp t;
        Bump b(chunk);
        const size_t ntiles = (n + RS_TILE - 1) / RS_TILE;
        t.keys_alt = b.take<uint32_t>(n);
        t.vals_alt = b.take<uint32_t>(n);
        t.vals2_alt = b.take<uint32_t>(n);
        t.H = b.take<uint32_t>(ntiles * RS_MAX_RADIX);
        t.totals = b.take<uint32_t>(RS_MAX_RADIX);
        t.skip = b.take<uint32_t>(RS_MAX_PASSES * 2);
        t.bytes = b.total();
        return t;
    }
};

}  // namespace

size_t sort_temp_bytes(size_t n) { return SortTemp::carve(nullptr, n).bytes; }

int sort_pairs_ex(void *temp, size_t temp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
                  const uint32_t *win, uint32_t *wout, size_t n, int end_bit, bool allow_skip, const uint32_t **totals_out,
                  hipStream_t s, uint32_t *inv_out)
{
    if (totals_out) *totals_out = nullptr;
    if (n == 0) return 0;
    if (n >= (size_t)1 << 31) {
        set_error("sort_pairs: %zu pairs exceed the 2^31 index range", n);
        return R2_ERR_INVALID;
    }
    const SortTemp t = SortTemp::carve(reinterpret_cast<char *>(temp), n);
    if (t.bytes > temp_bytes) {
        set_error("sort_pairs: temp storage too small (%zu < %zu)", temp_bytes, t.bytes);
        return R2_ERR_INVALID;
    }
    const Plan plan = make_plan(end_bit);
    const uint32_t ntiles = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
    const int phase = plan.npass & 1;
    const bool has_w = win != nullptr;
    Buffers buf{kin, vin, win, t.keys_alt, t.vals_alt, t.vals2_alt, kout, vout, wout};
    for (int p = 0; p < plan.npass; ++p) {
        const int bits = plan.bits[p], radix = 1 << bits;
        rs_upsweep_kernel<<<dim3(ntiles), dim3(RS_THREADS), 0, s>>>(buf, (uint32_t)n, plan.shift[p], bits, p, phase, t.skip,
                                                                     t.H, p == 0 ? t.skip : nullptr);
        rs_scan_kernel<<<dim3((radix + RS_SCAN_DIGITS - 1) / RS_SCAN_DIGITS), dim3(RS_SCAN_THREADS), 0, s>>>(
            t.H, ntiles, bits, (uint32_t)n, p, allow_skip ? 1 : 0, t.skip, t.totals);
        // per-wave histograms + digit offsets (+ for narrow digits: regrouped tile starts and the staged tile itself)
        const size_t lds = ((size_t)(RS_WAVES + 1) * pidx((uint32_t)radix) +
                            (bits <= 8 ? pidx((uint32_t)radix) + 3 * (size_t)RS_TILE : 0)) * sizeof(uint32_t);
        if (inv_out && has_w && !allow_skip && p == plan.npass - 1)   // last pass: inverse permutation instead of the values
            rs_downsweep_kernel<true, false, true><<<dim3(ntiles), dim3(RS_THREADS), lds, s>>>(
                buf, (uint32_t)n, plan.shift[p], bits, p, phase, t.skip, t.H, t.totals, inv_out, WorkListOut{});
        else if (has_w)
            rs_downsweep_kernel<true, false><<<dim3(ntiles), dim3(RS_THREADS), lds, s>>>(
                buf, (uint32_t)n, plan.shift[p], bits, p, phase, t.skip, t.H, t.totals, nullptr, WorkListOut{});
        else
            rs_downsweep_kernel<false, false><<<dim3(ntiles), dim3(RS_THREADS), lds, s>>>(
                buf, (uint32_t)n, plan.shift[p], bits, p, phase, t.skip, t.H, t.totals, nullptr, WorkListOut{});
    }
    if (allow_skip) {
        const uint32_t grid = (uint32_t)std::min<size_t>(1024, (n + RS_THREADS - 1) / RS_THREADS);
        if (has_w) rs_finalize_kernel<true><<<dim3(grid), dim3(RS_THREADS), 0, s>>>(buf, (uint32_t)n, plan.npass, phase, t.skip);
        else rs_finalize_kernel<false><<<dim3(grid), dim3(RS_THREADS), 0, s>>>(buf, (uint32_t)n, plan.npass, phase, t.skip);
    }
    if (totals_out && plan.npass == 1) *totals_out = t.totals;   // bucket sizes of the (only) digit
    R2_HIP_TRY(hipGetLastError());
    return 0;
}

bool sort_is_single_pass(int end_bit) { return make_plan(end_bit).npass == 1; }

// keys per workgroup of the single-pass tile sort: its two big kernels run one workgroup per key tile and a CU hosts them one after
// the other in practice, so the kernel lasts ceil(tiles / CUs) rounds of a workgroup's life (~ its keys); pick the tile size with
// the least rounds x keys (1.16 M instances: 284 tiles of 4096 = two rounds, 252 tiles of 4608 = one)
static int pick_ipt(size_t n)
{
    const int cus = device_cu_count();
    int ipt = RS_IPT;
    static const int cand[] = { 8, 9, 10, 12 };
    size_t best = ~(size_t)0;
    for (int c : cand) {
        const size_t tiles = (n + (size_t)c * RS_THREADS - 1) / ((size_t)c * RS_THREADS);
        const size_t cost = ((tiles + (size_t)cus - 1) / (size_t)cus) * (size_t)c;
        if (cost < best) { best = cost; ipt = c; }
    }
    return ipt;
}

// Where a producer that already holds the keys can leave the per-tile digit histograms itself (rs_upsweep's output): row t of
// H = counts of the `radix` digit values among keys [t * tile_keys, (t + 1) * tile_keys); skip[0] must be cleared.
bool tile_sort_plan(void *temp, size_t temp_bytes, size_t n, int end_bit, TileSortPlan *out)
{
    const Plan plan = make_plan(end_bit);
    if (n == 0 || plan.npass != 1 || n >= (size_t)1 << 31) return false;
    const SortTemp t = SortTemp::carve(reinterpret_cast<char *>(temp), n);
    if (t.bytes > temp_bytes) return false;
    const int ipt = pick_ipt(n);
    out->tile_keys = (uint32_t)(ipt * RS_THREADS);
    out->ntiles = (uint32_t)((n + out->tile_keys - 1) / out->tile_keys);
    out->bits = plan.bits[0];
    out->H = t.H;
    out->skip = t.skip;
    return true;
}

// Single-pass stable sort of n instances by tile id (end_bit <= 12 bits): ids_out[pos] = ids[index] (scattered),
// inv_out[index] = pos (coalesced), *counts_out = device pointer to the per-tile instance counts.
int sort_by_tile_single_pass(void *temp, size_t temp_bytes, const uint32_t *tiles, const uint32_t *ids, uint32_t *ids_out,
                             uint32_t *inv_out, size_t n, int end_bit, const uint32_t **counts_out, hipStream_t s,
                             const WorkListOut *work_out, bool hist_ready)
{
    if (counts_out) *counts_out = nullptr;
    if (n == 0) return 0;
    const Plan plan = make_plan(end_bit);
    if (plan.npass != 1 || n >= (size_t)1 << 31) {
        set_error("sort_by_tile_single_pass: %d key bits / %zu instances not supported", end_bit, n);
        return R2_ERR_INVALID;
    }
    const SortTemp t = SortTemp::carve(reinterpret_cast<char *>(temp), n);
    if (t.bytes > temp_bytes) {
        set_error("sort_by_tile_single_pass: temp storage too small (%zu < %zu)", temp_bytes, t.bytes);
        return R2_ERR_INVALID;
    }
    const int ipt = pick_ipt(n);
    const uint32_t ntiles = (uint32_t)((n + (size_t)ipt * RS_THREADS - 1) / ((size_t)ipt * RS_THREADS));
    const int bits = plan.bits[0], radix = 1 << bits, phase = 1;
    Buffers buf{tiles, nullptr, ids, nullptr, nullptr, nullptr, nullptr, nullptr, ids_out};
    if (!hist_ready)   // (else the producer of the keys built the histograms: tile_sort_plan)
        rs_upsweep_kernel<<<dim3(ntiles), dim3(RS_THREADS), 0, s>>>(buf, (uint32_t)n, 0, bits, 0, phase, t.skip, t.H, t.skip, ipt);
    rs_scan_kernel<<<dim3((radix + RS_SCAN_DIGITS - 1) / RS_SCAN_DIGITS), dim3(RS_SCAN_THREADS), 0, s>>>(
        t.H, ntiles, bits, (uint32_t)n, 0, 0, t.skip, t.totals);
    const size_t lds = (size_t)(RS_WAVES + 1) * pidx((uint32_t)radix) * sizeof(uint32_t);
    const WorkListOut wo = work_out ? *work_out : WorkListOut{};
    const dim3 grid(ntiles + (wo.ranges ? 1u : 0u));   // + the workgroup that builds tile ranges and the work list
    switch (ipt) {
    case 9: rs_downsweep_kernel<true, true, false, 9><<<grid, dim3(RS_THREADS), lds, s>>>(buf, (uint32_t)n, 0, bits, 0, phase, t.skip, t.H, t.totals, inv_out, wo); break;
    case 10: rs_downsweep_kernel<true, true, false, 10><<<grid, dim3(RS_THREADS), lds, s>>>(buf, (uint32_t)n, 0, bits, 0, phase, t.skip, t.H, t.totals, inv_out, wo); break;
    case 12: rs_downsweep_kernel<true, true, false, 12><<<grid, dim3(RS_THREADS), lds, s>>>(buf, (uint32_t)n, 0, bits, 0, phase, t.skip, t.H, t.totals, inv_out, wo); break;
    default: rs_downsweep_kernel<true, true, false, 8><<<grid, dim3(RS_THREADS), lds, s>>>(buf, (uint32_t)n, 0, bits, 0, phase, t.skip, t.H, t.totals, inv_out, wo); break;
    }
    if (counts_out) *counts_out = t.totals;
    R2_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace r2
