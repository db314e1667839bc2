// binning.hip -- prefix sum, tile-range detection, work-list construction and the stage profiler shared by
// the rasterizer and the voxelizer.  Replaces cub::DeviceScan::InclusiveSum / identifyTileRanges of the
// reference (RAS/rasterizer_impl.cu:116-138,275,308-316); the sort lives in radix_sort.hip.
#include "r2_common.hpp"
#include <chrono>
#include <mutex>
#include <cstdlib>
#include <cstring>
#include <stdarg.h>
#include <vector>
#include <string.h>

namespace r2 {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char *get_error() { return g_err; }

// ---- stage timers: event pairs recorded on the caller's stream, resolved lazily in r2_profile_read
int g_profile_mask_on = 0;
static unsigned long long g_profile_mask = 0;
struct Pending { int stage; hipEvent_t a, b; };
static std::vector<Pending> g_pending;
static std::vector<hipEvent_t> g_pool;
static hipEvent_t g_open[ST_COUNT];
static double g_ms[ST_COUNT];
static long long g_cnt[ST_COUNT];
static const char *const g_stage_names[ST_COUNT] = {
    "raster.preprocess", "raster.scan", "raster.duplicate", "raster.sort", "raster.ranges", "raster.render_fwd",
    "raster.render_bwd", "raster.geom_bwd", "voxel.preprocess", "voxel.scan", "voxel.duplicate", "voxel.sort",
    "voxel.ranges", "voxel.render_fwd", "voxel.render_bwd", "voxel.geom_bwd", "knn.dist2", "raster.depth_sort",
    "voxel.depth_sort"};

static hipEvent_t pool_get()
{
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void stage_begin(int stage, hipStream_t s)
{
    if (!((g_profile_mask >> stage) & 1ull)) return;
    hipEvent_t e = pool_get();
    (void)hipEventRecord(e, s);
    g_open[stage] = e;
}
void stage_end(int stage, hipStream_t s)
{
    if (!((g_profile_mask >> stage) & 1ull)) return;
    hipEvent_t e = pool_get();
    (void)hipEventRecord(e, s);
    g_pending.push_back({stage, g_open[stage], e});
}

static double g_sync_wait_us = 0.0;
static long long g_sync_calls = 0;
// host time of the forward passes around their synchronisation point (r2_profile_host)
static double g_pre_sync_us = 0.0, g_post_sync_us = 0.0;
static long long g_fwd_calls = 0;
static thread_local std::chrono::steady_clock::time_point g_fwd_t0, g_fwd_t1;
void host_mark_forward_begin() { g_fwd_t0 = std::chrono::steady_clock::now(); }
void host_mark_wait_begin() { g_pre_sync_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - g_fwd_t0).count(); }
void host_mark_wait_end() { g_fwd_t1 = std::chrono::steady_clock::now(); }
void host_mark_forward_end()
{
    g_post_sync_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - g_fwd_t1).count();
    g_fwd_calls += 1;
}

// pinned destination (pageable ones are staged and synchronised by the runtime) and a busy-wait on an event: the GPU is
// idle until the host has seen these words and launched the rest of the forward pass, so wake-up latency is on the
// critical path -- a blocking hipStreamSynchronize may sleep on an interrupt (tens of microseconds)
// pinned host words of the calling thread (the device -> host read-back and the mailbox below): released when the thread exits
struct PinnedWords {
    uint32_t *p = nullptr;
    void release() { if (p) { if (hipHostFree(p) != hipSuccess) (void)hipGetLastError(); p = nullptr; } }
    ~PinnedWords() { release(); }
};
static thread_local PinnedWords g_pinned_holder, g_mailbox_holder;
#define g_pinned g_pinned_holder.p
// one event per device: an event can only be recorded on a stream of the device it was created on, and one host thread may
// drive several GPUs (the caller makes the stream's device current, like every HIP API that takes a stream)
constexpr int MAX_DEVICES = R2_MAX_DEVICES;
static thread_local hipEvent_t g_read_evs[MAX_DEVICES] = {};
static thread_local hipEvent_t g_read_ev = nullptr;   // the event of the read in flight

int current_device_slot()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev % MAX_DEVICES;
}

// A kernel that needs more than the default 64 KB of dynamic LDS has to be told so once PER DEVICE (a host thread may drive several
// GPUs): state[] is the call site's table (0 = not asked yet, 1 = granted, -1 = refused), indexed by current_device_slot().
bool allow_dynamic_lds(const void *kernel, int bytes, signed char *state)
{
    signed char &st = state[current_device_slot()];   // written with the same value by whoever gets here first: a benign race
    if (st == 0) {
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess) st = 1;
        else { (void)hipGetLastError(); st = -1; }
    }
    return st > 0;
}

size_t device_lds_optin_bytes()
{
    static size_t lim[MAX_DEVICES] = {};   // written once per device with the same value
    const int slot = current_device_slot();
    if (lim[slot] == 0) {
        int dev = 0, a = 0, b = 0;
        if (hipGetDevice(&dev) != hipSuccess) (void)hipGetLastError();
        // runtimes differ in which of the two attributes carries the opt-in limit (gfx950: 160 KB): take the larger
        if (hipDeviceGetAttribute(&a, hipDeviceAttributeSharedMemPerBlockOptin, dev) != hipSuccess) { (void)hipGetLastError(); a = 0; }
        if (hipDeviceGetAttribute(&b, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) { (void)hipGetLastError(); b = 0; }
        const int n = std::max(std::max(a, b), 64 * 1024);   // 64 KB: what every device grants without asking
        lim[slot] = (size_t)n;
    }
    return lim[slot];
}

int device_cu_count()
{
    static int cus[MAX_DEVICES] = {};   // written once per device with the same value: a benign race at worst
    const int slot = current_device_slot();
    if (cus[slot] == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        cus[slot] = n;
    }
    return cus[slot];
}

int read_host_words_begin(const uint32_t *dev_words, int n, hipStream_t s)
{
    if (n < 1 || n > 16) {
        set_error("read_host_words: %d words", n);
        return R2_ERR_INVALID;
    }
    if (!g_pinned) R2_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&g_pinned), 64, hipHostMallocPortable));
    hipEvent_t &ev = g_read_evs[current_device_slot()];
    if (!ev) R2_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    g_read_ev = ev;
    R2_HIP_TRY(hipMemcpyAsync(g_pinned, dev_words, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    R2_HIP_TRY(hipEventRecord(g_read_ev, s));
    return 0;
}

int read_host_words_wait(uint32_t *out, int n)
{
    hipError_t q;
    host_mark_wait_begin();
    const auto t0 = std::chrono::steady_clock::now();
    while ((q = hipEventQuery(g_read_ev)) == hipErrorNotReady) __builtin_ia32_pause();
    g_sync_wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    g_sync_calls += 1;
    host_mark_wait_end();
    if (q != hipSuccess) {
        set_error("read_host_words: %s", hipGetErrorString(q));
        return -(int)q;
    }
    for (int i = 0; i < n; ++i) out[i] = g_pinned[i];
    return 0;
}

// Zero-copy variant for the hinted forward path: the kernel that produces the last of the words stores them, then a
// sequence number (release, system scope), straight into pinned host memory, and the host spins on the sequence number.
// No copy kernel (4 us of stream time on this runtime + a ~5 us signalling gap behind it) and no event.
#define g_mailbox g_mailbox_holder.p   // [16]: words 0..14, sequence number at 15
static thread_local uint32_t g_mailbox_seq = 0;

int host_mailbox_arm(uint32_t **mailbox, uint32_t *seq)
{
    if (!g_mailbox) {
        R2_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&g_mailbox), 64,
                                 hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
        memset(g_mailbox, 0, 64);
    }
    if (++g_mailbox_seq == 0u) ++g_mailbox_seq;   // 0 = never written
    *mailbox = g_mailbox;
    *seq = g_mailbox_seq;
    return 0;
}

void host_words_release() { g_pinned_holder.release(); g_mailbox_holder.release(); }

static int mailbox_spin(uint32_t *words, uint32_t seq, uint32_t *out, int n, hipStream_t s);
int host_mailbox_wait(uint32_t seq, uint32_t *out, int n, hipStream_t s) { return mailbox_spin(g_mailbox, seq, out, n, s); }

static int mailbox_spin(uint32_t *words, uint32_t seq, uint32_t *out, int n, hipStream_t s)
{
    host_mark_wait_begin();
    const auto t0 = std::chrono::steady_clock::now();
    volatile uint32_t *mb = words;
    unsigned spins = 0;
    bool drained = false;
    static const double timeout_s = [] { const char *e = getenv("R2_SYNC_TIMEOUT_S"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 30.0; }();
    while (__atomic_load_n(&words[15], __ATOMIC_ACQUIRE) != seq) {
        __builtin_ia32_pause();
        if ((++spins & 0x3FFFu) == 0u) {   // the producing kernel never ran?  (launch failure: do not spin forever)
            // a hung GPU or a stream blocked on something that never happens: give up after a wall-clock limit instead of
            // spinning forever (R2_SYNC_TIMEOUT_S, default 30 s); kernels queued so far may still write the mailbox later,
            // which is harmless -- the next call uses a new sequence number
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
                set_error("host_mailbox_wait: no control words from the GPU after %.0f s (hung device or blocked stream)", timeout_s);
                return R2_ERR_INVALID;
            }
            const hipError_t q = hipStreamQuery(s);
            if (q != hipSuccess && q != hipErrorNotReady) {
                set_error("host_mailbox_wait: %s", hipGetErrorString(q));
                return -(int)q;
            }
            if (q == hipSuccess) {
                if (drained) {
                    set_error("host_mailbox_wait: stream drained without the forward's control words arriving");
                    return R2_ERR_INVALID;
                }
                drained = true;
            }
        }
    }
    g_sync_wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    g_sync_calls += 1;
    host_mark_wait_end();
    for (int i = 0; i < n; ++i) out[i] = mb[i];
    return 0;
}

// ---- deferred num_rendered (round 6; r2_defer_count_control).  A forward that does not wait for its control words leaves them in a
// SLOT of a small process-wide pool of pinned words and returns a token naming the slot; whoever calls the matching backward -- with
// torch that is the autograd engine's thread, not the forward's -- resolves the token there: by then the forward's second kernel
// has long posted the words, so the wait that cost the forward ~95 of its 170 us of host time is gone, not moved.
// Slots are handed out by one mutex-protected table; a slot whose forward has completed but whose backward never came (rendering
// under no_grad) is recycled, least recently used first.
namespace {
constexpr int DEFER_SLOTS = 64;
struct DeferSlot { uint32_t seq = 0; uint32_t cap = 0; bool busy = false; unsigned long long used = 0; };
std::mutex g_defer_mu;
uint32_t *g_defer_words = nullptr;   // [DEFER_SLOTS][16] pinned, never freed (process lifetime: tokens may outlive any thread)
DeferSlot g_defer[DEFER_SLOTS];
uint32_t g_defer_seq = 0;
unsigned long long g_defer_tick = 0;
}  // namespace

int defer_acquire(uint32_t **mailbox, uint32_t *seq, uint32_t cap)
{
    std::lock_guard<std::mutex> lk(g_defer_mu);
    if (!g_defer_words) {
        if (hipHostMalloc(reinterpret_cast<void **>(&g_defer_words), DEFER_SLOTS * 64,
                          hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
            (void)hipGetLastError();
            g_defer_words = nullptr;
            return -1;
        }
        memset(g_defer_words, 0, DEFER_SLOTS * 64);
    }
    int pick = -1;
    for (int i = 0; i < DEFER_SLOTS && pick < 0; ++i)
        if (!g_defer[i].busy) pick = i;
    if (pick < 0) {   // every slot is waiting for a backward: recycle the oldest one whose forward has posted its words
        for (int i = 0; i < DEFER_SLOTS; ++i)
            if (__atomic_load_n(&g_defer_words[i * 16 + 15], __ATOMIC_ACQUIRE) == g_defer[i].seq &&
                (pick < 0 || g_defer[i].used < g_defer[pick].used))
                pick = i;
        if (pick < 0) return -1;   // none: the caller falls back to the waiting forward
    }
    if (++g_defer_seq == 0u) ++g_defer_seq;
    g_defer[pick] = DeferSlot{g_defer_seq, cap, true, ++g_defer_tick};
    *mailbox = g_defer_words + pick * 16;
    *seq = g_defer_seq;
    return DEFER_TOKEN_FLAG | (pick << 16) | (int)(g_defer_seq & 0xFFFFu);
}

// -> 0 and the words, or an error: the token is stale (its slot was recycled), or the device never posted
int defer_resolve(int token, uint32_t *out, int n, uint32_t *cap, hipStream_t s, bool release)
{
    const int slot = (token >> 16) & (DEFER_SLOTS - 1);
    uint32_t seq = 0;
    {
        std::lock_guard<std::mutex> lk(g_defer_mu);
        if (!g_defer_words || !g_defer[slot].busy || (g_defer[slot].seq & 0xFFFFu) != (uint32_t)(token & 0xFFFF)) {
            set_error("deferred num_rendered: stale token (the forward's slot was recycled: more than %d forwards without a backward)",
                      DEFER_SLOTS);
            return R2_ERR_INVALID;
        }
        seq = g_defer[slot].seq;
        if (cap) *cap = g_defer[slot].cap;
    }
    const int rc = mailbox_spin(g_defer_words + slot * 16, seq, out, n, s);
    if (release) {
        std::lock_guard<std::mutex> lk(g_defer_mu);
        if (g_defer[slot].seq == seq) g_defer[slot].busy = false;
    }
    return rc;
}

// non-blocking: have the words of this token arrived?  (the forward's own thread keeps its predictions current with it)
bool defer_peek(int token, uint32_t *out, int n)
{
    const int slot = (token >> 16) & (DEFER_SLOTS - 1);
    std::lock_guard<std::mutex> lk(g_defer_mu);
    if (!g_defer_words || (g_defer[slot].seq & 0xFFFFu) != (uint32_t)(token & 0xFFFF)) return false;
    uint32_t *w = g_defer_words + slot * 16;
    if (__atomic_load_n(&w[15], __ATOMIC_ACQUIRE) != g_defer[slot].seq) return false;
    for (int i = 0; i < n; ++i) out[i] = w[i];
    return true;
}

int read_host_words(const uint32_t *dev_words, uint32_t *out, int n, hipStream_t s)
{
    const int rc = read_host_words_begin(dev_words, n, s);
    return rc ? rc : read_host_words_wait(out, n);
}

uint32_t higher_msb(uint32_t n)
{
    // smallest b with (n >> b) == 0, i.e. bit length of n (same value as the reference's
    // getHigherMsb binary search, RAS/rasterizer_impl.cu:35-50, for every n >= 1)
    uint32_t msb = 16, step = 16;
    while (step > 1) {
        step >>= 1;
        msb = (n >> msb) ? msb + step : msb - step;
    }
    if (n >> msb) msb++;
    return msb;
}

// ---- prefix sum (replaces cub::DeviceScan::InclusiveSum, RAS/rasterizer_impl.cu:275): two kernels, no
// inter-workgroup waiting.  K1: every workgroup reduces its 4096-element tile; K2: every workgroup adds up the
// partials of the tiles before it (<= a few hundred values) and scans its own tile.  `order` (optional) gathers the
// input: out[j] = sum_{i<=j} in[order[i]] -- the per-Gaussian tile counts visited in depth order.
constexpr int SC_THREADS = 1024;
constexpr int SC_IPT = 4;
constexpr int SC_TILE = SC_THREADS * SC_IPT;

__device__ __forceinline__ uint32_t block_reduce_1024(uint32_t v, uint32_t *sh /* [16] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < SC_THREADS / 64; ++w) t += sh[w];
    __syncthreads();
    return t;
}

__global__ void __launch_bounds__(SC_THREADS) scan_reduce_kernel(const uint32_t *__restrict__ in,
                                                                 const uint32_t *__restrict__ order, uint32_t n,
                                                                 uint32_t *__restrict__ partial)
{
    __shared__ uint32_t sh[SC_THREADS / 64];
    const uint32_t base = blockIdx.x * SC_TILE + threadIdx.x * SC_IPT;
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < SC_IPT; ++i) {
        const uint32_t j = base + i;
        if (j < n) v += in[order ? order[j] : j];
    }
    const uint32_t t = block_reduce_1024(v, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ void __launch_bounds__(SC_THREADS) scan_apply_kernel(const uint32_t *__restrict__ in,
                                                                const uint32_t *__restrict__ order, uint32_t n,
                                                                const uint32_t *__restrict__ partial,
                                                                uint32_t *__restrict__ out, uint32_t *__restrict__ total_out)
{
    __shared__ uint32_t sh[SC_THREADS / 64];
    __shared__ uint32_t wsum[SC_THREADS / 64];
    uint32_t pre = 0;
    for (uint32_t g = threadIdx.x; g < blockIdx.x; g += SC_THREADS) pre += partial[g];
    const uint32_t tile_base = block_reduce_1024(pre, sh);
    const uint32_t base = blockIdx.x * SC_TILE + threadIdx.x * SC_IPT;
    uint32_t x[SC_IPT], sum = 0;
#pragma unroll
    for (int i = 0; i < SC_IPT; ++i) {
        const uint32_t j = base + i;
        x[i] = j < n ? in[order ? order[j] : j] : 0u;
        sum += x[i];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t run = tile_base + incl - sum;
    for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
    for (int i = 0; i < SC_IPT; ++i) {
        run += x[i];
        if (base + i < n) {
            out[base + i] = run;
            if (total_out && base + i == n - 1) *total_out = run;   // grand total, next to the other host-read words
        }
    }
}

size_t scan_temp_bytes(int P) { return sizeof(uint32_t) * ((size_t)(P + SC_TILE - 1) / SC_TILE + 32); }
size_t scan_gather_temp_bytes(int P) { return scan_temp_bytes(P); }

int inclusive_scan_gather_u32(void *temp, size_t temp_bytes, const uint32_t *in, const uint32_t *order, uint32_t *out,
                              int P, hipStream_t s, uint32_t *total_out)
{
    if (P <= 0) return 0;
    if (temp_bytes < scan_temp_bytes(P)) {
        set_error("inclusive_scan: temp storage too small");
        return R2_ERR_INVALID;
    }
    const uint32_t tiles = (uint32_t)((P + SC_TILE - 1) / SC_TILE);
    uint32_t *partial = reinterpret_cast<uint32_t *>(temp);
    scan_reduce_kernel<<<dim3(tiles), dim3(SC_THREADS), 0, s>>>(in, order, (uint32_t)P, partial);
    scan_apply_kernel<<<dim3(tiles), dim3(SC_THREADS), 0, s>>>(in, order, (uint32_t)P, partial, out, total_out);
    R2_HIP_TRY(hipGetLastError());
    return 0;
}

int inclusive_scan_u32(void *temp, size_t temp_bytes, const uint32_t *in, uint32_t *out, int P, hipStream_t s)
{
    return inclusive_scan_gather_u32(temp, temp_bytes, in, nullptr, out, P, s, nullptr);
}

// One thread per sorted instance; a tile boundary writes the end of the previous tile's range and the
// start of the next one.  ranges must be zeroed first (tiles with no instance keep (0,0)).
__global__ void __launch_bounds__(256) tile_ranges_kernel(const uint32_t *__restrict__ tiles,
                                                          const uint32_t *__restrict__ perm,
                                                          const uint32_t *__restrict__ vals_unsorted,
                                                          uint32_t *__restrict__ point_list, uint32_t L,
                                                          uint2 *__restrict__ ranges)
{
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= L) return;
    if (point_list) point_list[idx] = vals_unsorted[perm[idx]];   // absent when the sort carried the ids itself
    const uint32_t cur = tiles[idx];
    if (idx == 0) ranges[cur].x = 0;
    else {
        const uint32_t prev = tiles[idx - 1];
        if (cur != prev) {
            ranges[prev].y = idx;
            ranges[cur].x = idx;
        }
    }
    if (idx == L - 1) ranges[cur].y = L;
}

int tile_ranges(const uint32_t *tiles_sorted, const uint32_t *perm, const uint32_t *vals_unsorted, uint32_t *point_list,
                size_t R, uint2 *ranges, size_t T, hipStream_t s, bool ranges_zeroed)
{
    if (!ranges_zeroed) R2_HIP_TRY(hipMemsetAsync(ranges, 0, T * sizeof(uint2), s));   // (else: an earlier kernel of the caller did it)
    if (R > 0)
        tile_ranges_kernel<<<dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s>>>(tiles_sorted, perm, vals_unsorted,
                                                                                  point_list, (uint32_t)R, ranges);
    return 0;
}

__global__ void __launch_bounds__(256) fill_tiles_kernel(const uint2 *__restrict__ ranges, uint32_t *__restrict__ tiles)
{
    const uint2 r = ranges[blockIdx.x];
    for (uint32_t k = r.x + threadIdx.x; k < r.y; k += 256) tiles[k] = blockIdx.x;
}
int fill_tiles_from_ranges(const uint2 *ranges, size_t T, uint32_t *tiles, hipStream_t s)
{
    if (T > 0) fill_tiles_kernel<<<dim3((unsigned)T), dim3(256), 0, s>>>(ranges, tiles);
    return 0;
}
// The tile lists are cut into work items of `chunk` instances for the render kernels (load balance).
// work list: tile t owns work items [chunk_base[t], chunk_base[t+1]).  One workgroup, T is small (<= 2^20).
__global__ void __launch_bounds__(1024) build_work_kernel(const uint2 *__restrict__ ranges, uint32_t T, uint32_t chunk,
                                                          uint32_t *__restrict__ chunk_base,
                                                          uint4 *__restrict__ work_tile, uint32_t min_len)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < T; base += 1024) {
        const uint32_t t = base + tid;
        uint32_t n = 0;
        if (t < T) {
            const uint2 r = ranges[t];
            const uint32_t ch = work_tile_chunk(chunk, r.y - r.x);
            n = (r.y - r.x) < min_len ? 0u : (r.y - r.x + ch - 1) / ch;
        }
        // inclusive scan inside the wave, then across the 16 waves
        uint32_t incl = n;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const uint32_t excl = carry + woff + incl - n;
        if (t < T) {
            chunk_base[t] = excl;
            const uint2 r = ranges[t];
            const uint32_t ch = work_tile_chunk(chunk, r.y - r.x);
            for (uint32_t j = 0; j < n; ++j)   // work descriptor: {tile, first instance, one past the last, items of the tile}
                work_tile[excl + j] = make_uint4(t, r.x + j * ch, min(r.y, r.x + (j + 1) * ch), n);
        }
        __syncthreads();
        if (tid == 1023) carry = excl + n;
        __syncthreads();
    }
    if (tid == 0) chunk_base[T] = carry;
}

// (Measured and left out, round 4: ONE workgroup for 32768 tiles as well, every thread owning 32 consecutive tiles, two passes over
// the L2-resident ranges with eight loads in flight -- one launch instead of four, but 45 us slower: 43 k stores through one CU.)
// many tiles (256^3 volume: 32768): the same in three parallel steps -- per-tile work item counts, their prefix sum
// (own scan above), then every tile writes its base and its work items
__global__ void __launch_bounds__(256) work_count_kernel(const uint2 *__restrict__ ranges, uint32_t T, uint32_t chunk,
                                                         uint32_t *__restrict__ nw, uint32_t min_len)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= T) return;
    const uint2 r = ranges[t];
    const uint32_t ch = work_tile_chunk(chunk, r.y - r.x);
    nw[t] = (r.y - r.x) < min_len ? 0u : (r.y - r.x + ch - 1) / ch;
}
__global__ void __launch_bounds__(256) work_fill_kernel(const uint2 *__restrict__ ranges, uint32_t chunk,
                                                        const uint32_t *__restrict__ nw, const uint32_t *__restrict__ incl,
                                                        uint32_t T, uint32_t *__restrict__ chunk_base,
                                                        uint4 *__restrict__ work_tile)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= T) return;
    const uint32_t n = nw[t], start = incl[t] - n;
    chunk_base[t] = start;
    const uint2 r = ranges[t];
    const uint32_t ch = work_tile_chunk(chunk, r.y - r.x);
    for (uint32_t j = 0; j < n; ++j)
        work_tile[start + j] = make_uint4(t, r.x + j * ch, min(r.y, r.x + (j + 1) * ch), n);
    if (t == T - 1) chunk_base[T] = incl[t];
}

// Round 4: the same in TWO launches -- the scan's two kernels compute the per-tile counts from the ranges themselves, and the second
// one writes bases and work items on the way (count + scan-reduce, scan-apply + fill): two ~5 us launches less per 256^3 query.
__device__ __forceinline__ uint32_t work_items_of(const uint2 r, uint32_t chunk, uint32_t min_len)
{
    const uint32_t len = r.y - r.x, ch = work_tile_chunk(chunk, len);
    return len < min_len ? 0u : (len + ch - 1) / ch;
}
__global__ void __launch_bounds__(SC_THREADS) work_reduce_kernel(const uint2 *__restrict__ ranges, uint32_t T, uint32_t chunk,
                                                                 uint32_t min_len, uint32_t *__restrict__ partial)
{
    __shared__ uint32_t sh[SC_THREADS / 64];
    const uint32_t base = blockIdx.x * SC_TILE + threadIdx.x * SC_IPT;
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < SC_IPT; ++i)
        if (base + i < T) v += work_items_of(ranges[base + i], chunk, min_len);
    const uint32_t t = block_reduce_1024(v, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}
__global__ void __launch_bounds__(SC_THREADS) work_apply_fill_kernel(const uint2 *__restrict__ ranges, uint32_t T, uint32_t chunk,
                                                                     uint32_t min_len, const uint32_t *__restrict__ partial,
                                                                     uint32_t *__restrict__ chunk_base, uint4 *__restrict__ work_tile)
{
    __shared__ uint32_t sh[SC_THREADS / 64];
    __shared__ uint32_t wsum[SC_THREADS / 64];
    uint32_t pre = 0;
    for (uint32_t g = threadIdx.x; g < blockIdx.x; g += SC_THREADS) pre += partial[g];
    const uint32_t tile_base = block_reduce_1024(pre, sh);
    const uint32_t base = blockIdx.x * SC_TILE + threadIdx.x * SC_IPT;
    uint2 r[SC_IPT];
    uint32_t x[SC_IPT], sum = 0;
#pragma unroll
    for (int i = 0; i < SC_IPT; ++i) {
        r[i] = base + i < T ? ranges[base + i] : make_uint2(0u, 0u);
        x[i] = base + i < T ? work_items_of(r[i], chunk, min_len) : 0u;
        sum += x[i];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t run = tile_base + incl - sum;
    for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
    for (int i = 0; i < SC_IPT; ++i) {
        const uint32_t t = base + i;
        if (t < T) {
            chunk_base[t] = run;
            const uint32_t ch = work_tile_chunk(chunk, r[i].y - r[i].x);
            for (uint32_t j = 0; j < x[i]; ++j)   // work descriptor: {tile, first instance, one past the last, items of the tile}
                work_tile[run + j] = make_uint4(t, r[i].x + j * ch, min(r[i].y, r[i].x + (j + 1) * ch), x[i]);
            run += x[i];
            if (t == T - 1) chunk_base[T] = run;
        }
    }
}

// the second half of the two-launch construction alone, for a caller whose own kernels have left the per-block work item counts
// (partial[b] = work items of tiles [b * build_work_block_tiles(), ...)) -- voxel_sticks.hip: its sort kernel knows every tile's length
uint32_t build_work_block_tiles() { return (uint32_t)SC_TILE; }
void launch_build_work_from_partials(const uint2 *ranges, uint32_t T, uint32_t chunk, uint32_t *chunk_base, uint4 *work_tile,
                                     const uint32_t *partial, hipStream_t s, uint32_t min_len)
{
    const uint32_t tiles = (T + SC_TILE - 1) / SC_TILE;
    work_apply_fill_kernel<<<dim3(tiles), dim3(SC_THREADS), 0, s>>>(ranges, T, chunk, min_len, partial, chunk_base, work_tile);
}

size_t build_work_temp_bytes(size_t T) { return T > 4096 ? sizeof(uint32_t) * 2 * T + scan_temp_bytes((int)T) + 256 : 0; }

void launch_build_work(const uint2 *ranges, uint32_t T, uint32_t chunk, uint32_t *chunk_base, uint4 *work_tile,
                       void *temp, hipStream_t s, uint32_t min_len)
{
    if (T <= 4096 || temp == nullptr) {
        build_work_kernel<<<dim3(1), dim3(1024), 0, s>>>(ranges, T, chunk, chunk_base, work_tile, min_len);
        return;
    }
    static const bool two = [] { const char *e = getenv("R2_WORK_TWO"); return !(e && e[0] == '0'); }();
    if (two) {
        uint32_t *partial = reinterpret_cast<uint32_t *>(temp);   // (T + SC_TILE - 1) / SC_TILE words of the 2 T + ... reserved
        const uint32_t tiles = (T + SC_TILE - 1) / SC_TILE;
        work_reduce_kernel<<<dim3(tiles), dim3(SC_THREADS), 0, s>>>(ranges, T, chunk, min_len, partial);
        work_apply_fill_kernel<<<dim3(tiles), dim3(SC_THREADS), 0, s>>>(ranges, T, chunk, min_len, partial, chunk_base, work_tile);
        return;
    }
    uint32_t *nw = reinterpret_cast<uint32_t *>(temp), *incl = nw + T;
    void *scan_tmp = incl + T;
    work_count_kernel<<<dim3((T + 255) / 256), dim3(256), 0, s>>>(ranges, T, chunk, nw, min_len);
    (void)inclusive_scan_u32(scan_tmp, scan_temp_bytes((int)T), nw, incl, (int)T, s);
    work_fill_kernel<<<dim3((T + 255) / 256), dim3(256), 0, s>>>(ranges, chunk, nw, incl, T, chunk_base, work_tile);
}

// Single-pass tile sort: the sort's digit totals are the per-tile instance counts, so the tile ranges
// (identifyTileRanges, RAS/rasterizer_impl.cu:116-138) are their exclusive scan -- computed here together with the
// forward work list, by one workgroup (T <= 4096).
__global__ void __launch_bounds__(1024) ranges_and_work_kernel(const uint32_t *__restrict__ counts, WorkListOut wo)
{
    ranges_and_work_block<1024>(counts, wo);
}

void launch_ranges_and_work(const uint32_t *tile_counts, uint32_t T, uint32_t chunk, uint2 *ranges, uint32_t *chunk_base,
                            uint4 *work_tile, hipStream_t s, uint32_t min_len)
{
    ranges_and_work_kernel<<<dim3(1), dim3(1024), 0, s>>>(tile_counts, WorkListOut{ranges, chunk_base, work_tile, T, chunk, nullptr, min_len});
}

}  // namespace r2

extern "C" const char *r2_last_error(void) { return r2::get_error(); }

extern "C" int r2_abi_version(void) { return R2_ABI_VERSION; }

extern "C" int r2_sync_wait_stats(double *total_us, long long *calls, int reset)
{
    if (total_us) *total_us = r2::g_sync_wait_us;
    if (calls) *calls = r2::g_sync_calls;
    if (reset) { r2::g_sync_wait_us = 0.0; r2::g_sync_calls = 0; }
    return 0;
}

extern "C" int r2_profile_host(double *pre_sync_us, double *post_sync_us, long long *calls, int reset)
{
    if (pre_sync_us) *pre_sync_us = r2::g_pre_sync_us;
    if (post_sync_us) *post_sync_us = r2::g_post_sync_us;
    if (calls) *calls = r2::g_fwd_calls;
    if (reset) { r2::g_pre_sync_us = r2::g_post_sync_us = 0.0; r2::g_fwd_calls = 0; }
    return 0;
}

extern "C" void r2_profile_enable(unsigned long long stage_mask)
{
    r2::g_profile_mask = stage_mask;
    r2::g_profile_mask_on = stage_mask != 0;
}
extern "C" int r2_profile_stage_count(void) { return r2::ST_COUNT; }
extern "C" const char *r2_profile_stage_name(int stage)
{
    return (stage >= 0 && stage < r2::ST_COUNT) ? r2::g_stage_names[stage] : "";
}
extern "C" int r2_profile_read(double *total_ms, long long *counts, int reset)
{
    using namespace r2;
    for (auto &p : g_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            g_ms[p.stage] += ms;
            g_cnt[p.stage] += 1;
        }
        g_pool.push_back(p.a);
        g_pool.push_back(p.b);
    }
    g_pending.clear();
    for (int i = 0; i < ST_COUNT; ++i) {
        if (total_ms) total_ms[i] = g_ms[i];
        if (counts) counts[i] = g_cnt[i];
    }
    if (reset) { memset(g_ms, 0, sizeof(g_ms)); memset(g_cnt, 0, sizeof(g_cnt)); }
    return ST_COUNT;
}
