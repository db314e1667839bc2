// knn.hip -- distCUDA2 of simple-knn: for every point the mean of the 3 smallest squared distances to the
// OTHER points (self excluded by index).  Call site: r2_gaussian/gaussian/gaussian_model.py:145-150.
// The upstream submodule (gitlab.inria.fr/bkerbl/simple-knn) is not vendored in the reference; its
// Morton-box search is exact, so any exact 3-NN search computes the same function.
//
// Two exact searches: a uniform-grid one for P >= 4096 (below, round 4) and this exhaustive O(P^2) one, LDS-tiled, for small or
// degenerate inputs and as the A/B reference (R2_KNN_GRID=0): one lane owns one query point, the workgroup
// streams all points through a 1024-point LDS tile (float4-padded, coalesced b128 loads, wave-uniform
// LDS broadcast reads), 3-best kept in registers.  It runs once per training run (P = 50k: 2.5e9 pairs).
// Compiled with -ffp-contract=off so dx*dx+dy*dy+dz*dz rounds exactly like the oracle; the 3-best
// multiset does not depend on the visiting order, so the result is bit-exact.
#include "r2_common.hpp"
#include <float.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace r2 {

constexpr int KNN_TILE = 1024;

__global__ void __launch_bounds__(256) knn_dist2_kernel(int P, const float *__restrict__ pts, float *__restrict__ out)
{
    __shared__ float4 tile[KNN_TILE];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float rx = 0.f, ry = 0.f, rz = 0.f;
    if (i < P) {
        rx = pts[3 * i];
        ry = pts[3 * i + 1];
        rz = pts[3 * i + 2];
    }
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    for (int base = 0; base < P; base += KNN_TILE) {
        __syncthreads();
        for (int t = threadIdx.x; t < KNN_TILE; t += 256) {
            const int j = base + t;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < P) v = make_float4(pts[3 * j], pts[3 * j + 1], pts[3 * j + 2], 0.f);
            tile[t] = v;
        }
        __syncthreads();
        const int n = min(KNN_TILE, P - base);
        for (int t = 0; t < n; ++t) {
            const float4 v = tile[t];
            const float dx = v.x - rx, dy = v.y - ry, dz = v.z - rz;
            float d = dx * dx + dy * dy + dz * dz;
            if (base + t == i) d = FLT_MAX;   // self excluded by index, duplicates at distance 0 still count
            // sorted insertion into (b0 <= b1 <= b2)
            const float n2 = fminf(b2, fmaxf(b1, d));
            const float n1 = fminf(b1, fmaxf(b0, d));
            const float n0 = fminf(b0, d);
            b0 = n0; b1 = n1; b2 = n2;
        }
    }
    if (i < P) out[i] = (b0 + b1 + b2) / 3.0f;
}


// ---- uniform-grid search (round 4).  The exhaustive kernel above is 33 ms at 300k points and 313 ms at 1M (10^12 pairs); the
// same exact result comes out of a grid: points sorted by cell (~2 per cell), every query walks the cells around its own in
// Chebyshev rings and stops when its third-best distance is within the part of space the rings have covered completely.  The
// distances are formed like above (other - self, x^2 + y^2 + z^2, no contraction) and the 3-best multiset does not depend on
// the visiting order, so the result is bit-identical to the exhaustive search (tests/test_knn_gpu.py: both against the oracle
// and against each other, incl. duplicates, collinear and clustered clouds).
struct KnnGrid {
    float ox, oy, oz;   // origin = bounding-box minimum
    float inv_h, h;     // cell size (isotropic)
    int gx, gy, gz;
};

__device__ __forceinline__ int3 knn_cell(const KnnGrid &g, float x, float y, float z)
{
    // (float -> int of a non-negative value; clamped, so every point of the box lands in a cell whatever the rounding)
    int3 c;
    c.x = min(g.gx - 1, max(0, (int)((x - g.ox) * g.inv_h)));
    c.y = min(g.gy - 1, max(0, (int)((y - g.oy) * g.inv_h)));
    c.z = min(g.gz - 1, max(0, (int)((z - g.oz) * g.inv_h)));
    return c;
}

__global__ void __launch_bounds__(256) knn_bbox_kernel(int P, const float *__restrict__ pts, float *__restrict__ partial /* [blocks][6] */)
{
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = pts[3 * i + a];
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
    }
    __shared__ float sm[4][6];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], d));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d));
        }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) { sm[wave][a] = lo[a]; sm[wave][3 + a] = hi[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        float v = sm[0][a];
        for (int w = 1; w < 4; ++w) v = a < 3 ? fminf(v, sm[w][a]) : fmaxf(v, sm[w][a]);
        partial[blockIdx.x * 6 + a] = v;
    }
}

__global__ void __launch_bounds__(256) knn_count_kernel(int P, const float *__restrict__ pts, KnnGrid g, uint32_t *__restrict__ count)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int3 c = knn_cell(g, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    atomicAdd(&count[((size_t)c.z * g.gy + c.y) * g.gx + c.x], 1u);
}

// the fullest cell (one word, zeroed with the counters): a clustered cloud defeats the grid -- every query of a cell holding
// O(P) points scans all of them, one lane per query -- and is handed to the exhaustive kernel instead
__global__ void __launch_bounds__(256) knn_maxcount_kernel(size_t cells, const uint32_t *__restrict__ count, uint32_t *__restrict__ out)
{
    uint32_t m = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < cells; i += (size_t)gridDim.x * 256) m = max(m, count[i]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// sorted[pos] = {x, y, z, bits(original index)}; start[c] = exclusive prefix of the counts (incl[c - 1]), cursor starts at 0
__global__ void __launch_bounds__(256) knn_scatter_kernel(int P, const float *__restrict__ pts, KnnGrid g, const uint32_t *__restrict__ incl,
                                                          uint32_t *__restrict__ cursor, float4 *__restrict__ sorted)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    const int3 c = knn_cell(g, x, y, z);
    const size_t cell = ((size_t)c.z * g.gy + c.y) * g.gx + c.x;
    const uint32_t base = cell ? incl[cell - 1] : 0u;
    const uint32_t pos = base + atomicAdd(&cursor[cell], 1u);
    sorted[pos] = make_float4(x, y, z, __uint_as_float((uint32_t)i));
}

__global__ void __launch_bounds__(256) knn_query_kernel(int P, KnnGrid g, const uint32_t *__restrict__ incl, const float4 *__restrict__ sorted,
                                                        float *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;   // a query per SORTED position: the lanes of a wave are neighbours in space
    if (i >= P) return;
    const float4 q = sorted[i];
    const uint32_t self = __float_as_uint(q.w);
    const int3 c = knn_cell(g, q.x, q.y, q.z);
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    const int rmax = max(g.gx, max(g.gy, g.gz));
    for (int r = 0; r <= rmax; ++r) {
        const int z0 = max(c.z - r, 0), z1 = min(c.z + r, g.gz - 1);
        const int y0 = max(c.y - r, 0), y1 = min(c.y + r, g.gy - 1);
        const int x0 = max(c.x - r, 0), x1 = min(c.x + r, g.gx - 1);
        for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y) {
                const bool shell_zy = (z == c.z - r) || (z == c.z + r) || (y == c.y - r) || (y == c.y + r);
                // on the shell in z or y: the whole x-run of cells (contiguous in the sorted array); else only the two end cells
                const size_t row = ((size_t)z * g.gy + y) * g.gx;
                if (shell_zy) {
                    const uint32_t j0 = (row + x0) ? incl[row + x0 - 1] : 0u, j1 = incl[row + x1];
                    for (uint32_t j = j0; j < j1; ++j) {
                        const float4 v = sorted[j];
                        const float dx = v.x - q.x, dy = v.y - q.y, dz = v.z - q.z;
                        float d = dx * dx + dy * dy + dz * dz;
                        if (__float_as_uint(v.w) == self) d = FLT_MAX;   // self excluded by index, duplicates at distance 0 count
                        const float n2 = fminf(b2, fmaxf(b1, d)), n1 = fminf(b1, fmaxf(b0, d)), n0 = fminf(b0, d);
                        b0 = n0; b1 = n1; b2 = n2;
                    }
                } else {
#pragma unroll
                    for (int side = 0; side < 2; ++side) {
                        const int x = side ? c.x + r : c.x - r;
                        if (x < 0 || x >= g.gx || (side && r == 0)) continue;
                        const uint32_t j0 = (row + x) ? incl[row + x - 1] : 0u, j1 = incl[row + x];
                        for (uint32_t j = j0; j < j1; ++j) {
                            const float4 v = sorted[j];
                            const float dx = v.x - q.x, dy = v.y - q.y, dz = v.z - q.z;
                            float d = dx * dx + dy * dy + dz * dz;
                            if (__float_as_uint(v.w) == self) d = FLT_MAX;
                            const float n2 = fminf(b2, fmaxf(b1, d)), n1 = fminf(b1, fmaxf(b0, d)), n0 = fminf(b0, d);
                            b0 = n0; b1 = n1; b2 = n2;
                        }
                    }
                }
            }
        // everything inside the block of cells [c - r, c + r]^3 has been seen (sides where the block already touches the grid's
        // border hold nothing).  How far away is whatever lies outside?  Derived from the cell function itself, not from the
        // nominal cell walls: a point p in a cell >= K along x satisfies fl(fl(p.x - ox) * inv_h) >= K, hence (two roundings,
        // each relative to its own result) p.x - ox >= K / inv_h * (1 - 3 * 2^-24) in exact arithmetic, wherever the cloud sits
        // and however many cells the axis has; a point in a cell <= K' satisfies p.x - ox < (K' + 1) / inv_h * (1 + 3 * 2^-24).
        // Evaluated in double (round 4 used the float walls shrunk by an ad hoc margin, which an elongated cloud of hundreds
        // of cells or a cloud far from the origin could exhaust: ADVICE r4).
        const bool all = x0 == 0 && y0 == 0 && z0 == 0 && x1 == g.gx - 1 && y1 == g.gy - 1 && z1 == g.gz - 1;
        if (all) break;
        constexpr double EPS3 = 3.0 / 16777216.0;
        const double ih = 1.0 / (double)g.inv_h;
        double reach = 1.0e300;
        if (c.x - r > 0) reach = fmin(reach, ((double)q.x - (double)g.ox) - (double)(c.x - r) * ih * (1.0 + EPS3));
        if (c.x + r < g.gx - 1) reach = fmin(reach, (double)(c.x + r + 1) * ih * (1.0 - EPS3) - ((double)q.x - (double)g.ox));
        if (c.y - r > 0) reach = fmin(reach, ((double)q.y - (double)g.oy) - (double)(c.y - r) * ih * (1.0 + EPS3));
        if (c.y + r < g.gy - 1) reach = fmin(reach, (double)(c.y + r + 1) * ih * (1.0 - EPS3) - ((double)q.y - (double)g.oy));
        if (c.z - r > 0) reach = fmin(reach, ((double)q.z - (double)g.oz) - (double)(c.z - r) * ih * (1.0 + EPS3));
        if (c.z + r < g.gz - 1) reach = fmin(reach, (double)(c.z + r + 1) * ih * (1.0 - EPS3) - ((double)q.z - (double)g.oz));
        // an unseen point's float distance (three squares and two sums, each rounded) is >= its true one * (1 - 2^-21): it cannot
        // displace b2, and the 3-best multiset is the exhaustive search's
        if (reach > 0.0 && (double)b2 <= reach * reach * (1.0 - 1.0e-6)) break;
    }
    out[self] = (b0 + b1 + b2) / 3.0f;
}

}  // namespace r2

// ---- host side.  The library does not allocate (include/r2hip.h): the grid search works inside a caller-provided workspace of
// r2_knn_workspace_bytes(P) bytes -- bounding-box partials, cell counters / cursors / prefix, scan temp, the points sorted by
// cell -- sized for the most cells the grid is allowed to have (2 P + 4096: the cell size is grown until it fits).
namespace {
constexpr int KNN_BBOX_BLOCKS = 256;
constexpr int KNN_GRID_MIN_POINTS = 4096;
inline size_t knn_align(size_t b) { return (b + 255) & ~(size_t)255; }
inline size_t knn_max_cells(int P) { return 2 * (size_t)P + 4096; }
struct KnnWs { size_t off_bbox, off_cnt, off_cur, off_max, off_incl, off_tmp, off_sorted, total, tmp_bytes; };
KnnWs knn_ws_layout(int P)
{
    KnnWs w;
    const size_t cb = knn_align(knn_max_cells(P) * 4);
    w.tmp_bytes = r2::scan_temp_bytes((int)knn_max_cells(P));
    w.off_bbox = 0;
    w.off_cnt = knn_align(sizeof(float) * 6 * KNN_BBOX_BLOCKS);
    w.off_cur = w.off_cnt + cb;
    w.off_max = w.off_cur + cb;           // (counters, cursors and this word are zeroed by one fill)
    w.off_incl = w.off_max + 256;
    w.off_tmp = w.off_incl + cb;
    w.off_sorted = w.off_tmp + knn_align(w.tmp_bytes);
    w.total = w.off_sorted + (size_t)P * sizeof(float4);
    return w;
}
}  // namespace

// -> 0, or a negative error code; 1 = not taken: run the exhaustive kernel (tiny inputs, degenerate boxes, clustered clouds)
static int knn_grid(int P, const float *points, float *out, char *ws, hipStream_t s)
{
    using namespace r2;
    constexpr int NOT_TAKEN = 1;
    static const bool on = [] { const char *e = getenv("R2_KNN_GRID"); return !(e && e[0] == '0'); }();
    if (!on || P < KNN_GRID_MIN_POINTS) return NOT_TAKEN;
    const KnnWs L = knn_ws_layout(P);
    // ---- bounding box (one small read-back: this call runs once per training run)
    const int nb = KNN_BBOX_BLOCKS;
    float *partial = reinterpret_cast<float *>(ws + L.off_bbox);
    knn_bbox_kernel<<<dim3(nb), dim3(256), 0, s>>>(P, points, partial);
    float hp[6 * nb];
    hipError_t e = hipMemcpyAsync(hp, partial, sizeof(hp), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { set_error("r2_knn_dist2: %s", hipGetErrorString(e)); return -(int)e; }
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int b = 0; b < nb; ++b)
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], hp[6 * b + a]);
            hi[a] = std::max(hi[a], hp[6 * b + 3 + a]);
        }
    double ext[3], vol = 1.0;
    for (int a = 0; a < 3; ++a) {
        if (!(lo[a] <= hi[a]) || !std::isfinite(lo[a]) || !std::isfinite(hi[a])) return NOT_TAKEN;   // NaN / inf coordinates
        ext[a] = (double)hi[a] - (double)lo[a];
    }
    const double emax = std::max(ext[0], std::max(ext[1], ext[2]));
    if (!(emax > 0.0)) return NOT_TAKEN;   // all points identical
    for (int a = 0; a < 3; ++a) vol *= std::max(ext[a], emax * 1e-3);   // flat clouds: a thin slab of cells
    // ~2 points per cell; at most 512 cells per axis and as many cells as the workspace was sized for
    double h = std::cbrt(vol / (0.5 * (double)P));
    h = std::max(h, emax / 512.0);
    KnnGrid g;
    size_t cells = 0;
    for (int it = 0; it < 64; ++it, h *= 1.1) {
        g.gx = std::max(1, (int)std::ceil(ext[0] / h)); g.gy = std::max(1, (int)std::ceil(ext[1] / h)); g.gz = std::max(1, (int)std::ceil(ext[2] / h));
        cells = (size_t)g.gx * g.gy * g.gz;
        if (cells <= knn_max_cells(P)) break;
    }
    if (cells > knn_max_cells(P)) return NOT_TAKEN;
    g.ox = lo[0]; g.oy = lo[1]; g.oz = lo[2];
    g.h = (float)h;
    g.inv_h = (float)(1.0 / h);
    uint32_t *cnt = reinterpret_cast<uint32_t *>(ws + L.off_cnt), *cur = reinterpret_cast<uint32_t *>(ws + L.off_cur),
             *mx = reinterpret_cast<uint32_t *>(ws + L.off_max), *incl = reinterpret_cast<uint32_t *>(ws + L.off_incl);
    float4 *sorted = reinterpret_cast<float4 *>(ws + L.off_sorted);
    e = hipMemsetAsync(ws + L.off_cnt, 0, L.off_incl - L.off_cnt, s);   // counts, cursors, the maximum
    if (e != hipSuccess) { set_error("r2_knn_dist2: %s", hipGetErrorString(e)); return -(int)e; }
    const unsigned blocks = (unsigned)((P + 255) / 256);
    knn_count_kernel<<<dim3(blocks), dim3(256), 0, s>>>(P, points, g, cnt);
    knn_maxcount_kernel<<<dim3((unsigned)std::min<size_t>((cells + 255) / 256, 1024)), dim3(256), 0, s>>>(cells, cnt, mx);
    uint32_t fullest = 0;
    e = hipMemcpyAsync(&fullest, mx, sizeof(fullest), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { set_error("r2_knn_dist2: %s", hipGetErrorString(e)); return -(int)e; }
    // a query scans ~27 cells, one lane per query; the exhaustive kernel pays P pairs per query but ~10x faster per pair
    if ((size_t)fullest > std::max<size_t>(256, (size_t)P / 256)) return NOT_TAKEN;
    const int rc = inclusive_scan_u32(ws + L.off_tmp, L.tmp_bytes, cnt, incl, (int)cells, s);
    if (rc) return rc;
    knn_scatter_kernel<<<dim3(blocks), dim3(256), 0, s>>>(P, points, g, incl, cur, sorted);
    knn_query_kernel<<<dim3(blocks), dim3(256), 0, s>>>(P, g, incl, sorted, out);
    e = hipGetLastError();
    if (e != hipSuccess) { set_error("r2_knn_dist2: %s", hipGetErrorString(e)); return -(int)e; }
    return 0;
}

extern "C" size_t r2_knn_workspace_bytes(int P)
{
    return P >= KNN_GRID_MIN_POINTS ? knn_ws_layout(P).total : 256;
}

extern "C" int r2_knn_dist2_ws(int P, const float *points, float *out, void *workspace, size_t workspace_bytes, void *stream)
{
    if (P == 0) return 0;
    if (P < 0 || !points || !out) {
        r2::set_error("r2_knn_dist2: invalid argument");
        return R2_ERR_INVALID;
    }
    { r2::StageScope t(r2::ST_KNN, (hipStream_t)stream);
    int g = 1;
    if (workspace != nullptr && workspace_bytes >= r2_knn_workspace_bytes(P))
        g = knn_grid(P, points, out, static_cast<char *>(workspace), (hipStream_t)stream);
    if (g <= 0) return g;
    r2::knn_dist2_kernel<<<dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(P, points, out); }
    R2_STAGE_CHECK(0, (hipStream_t)stream, "knn");
    return 0;
}

// convenience form (the reference's signature): obtains the workspace itself -- hipMalloc, and hipFree after waiting for the
// stream.  Callers that own an allocator (the torch boundary does) use r2_knn_workspace_bytes + r2_knn_dist2_ws.
extern "C" int r2_knn_dist2(int P, const float *points, float *out, void *stream)
{
    void *ws = nullptr;
    const size_t bytes = P > 0 ? r2_knn_workspace_bytes(P) : 0;
    if (P >= KNN_GRID_MIN_POINTS && hipMalloc(&ws, bytes) != hipSuccess) { (void)hipGetLastError(); ws = nullptr; }
    const int rc = r2_knn_dist2_ws(P, points, out, ws, ws ? bytes : 0, stream);
    if (ws) {
        (void)hipStreamSynchronize((hipStream_t)stream);
        (void)hipFree(ws);
    }
    return rc;
}
