// voxel_state.hpp -- private layout of the voxelizer's opaque state buffers (the reference's
// GeometryState / BinningState / ImageState, VOX/voxelizer_impl.h, VOX/voxelizer_impl.cu:130-169).
#pragma once
#include "r2_common.hpp"

namespace r2 {

constexpr uint32_t VOX_CHUNK = 1024;  // most instances of one tile list evaluated by one workgroup (load balance)
// Small problems (the training loop's 32^3 TV patch: 64 tiles, ~5e4 instances) are cut finer, so that the forward still
// launches a few workgroups per CU instead of ~one long-running workgroup per tile.  A pure function of the grid's y-z CROSS-SECTION
// (tiles): the backward and the state introspection re-derive the same layout, and -- round 6 -- so does every x-slab call on the
// same volume: where a tile's list is cut decides how its partial sums associate, so the unsharded query and its slabs must cut
// alike to be bit-identical (rounds 1-5 chose by the call's instance count R, which a slab cannot know of the full call).
// 32^3 patch: 128; 64^3: 256; 128^3: 512; 256^3: 1024.  This is the BASE chunk: the work-list builders are handed vox_work_chunk()
// below, under which a long list chooses a larger one for itself (work_tile_chunk) -- the round-5 rule gave a dense patch (the TV
// patch of the 331k trained cloud: 3 100 instances per tile) chunks of 256, the cross-section rule alone 128: 259 -> 295 us; with the
// list's own choice 239 us, a million Gaussians at 64^3 534 -> 413 us (profiles/r06e_voxel_adaptive_chunk_ab.txt).
__host__ __device__ inline uint32_t vox_chunk_for(int gy, int gz)
{
    const long long c = (long long)gy * gz;
    return c > 256 ? VOX_CHUNK : (c > 64 ? 512u : (c > 16 ? 256u : 128u));
}
// what the work-list builders are handed: the base chunk + "a long list chooses a larger one" (work_tile_chunk, r2_common.hpp) -- a
// function of the grid and of the list itself, so still the same in a slab call and in the unsharded one
__host__ __device__ inline uint32_t vox_work_chunk(int gy, int gz) { return vox_chunk_for(gy, gz) | WORK_CHUNK_ADAPT; }
constexpr int VPART_STRIDE = 12;      // floats per instance in the backward moment scratch (10 used)
constexpr float ALPHA_MIN_3D = 0.000001f;                 // VOX/forward.cu:293
constexpr float LOG2_ALPHA_MIN_3D = -19.931568569324174f;   // log2(1e-6)

// Row recurrence along z (see raster_render.hip: row_tier): G(c+1) = G(c) r(c), r(c+1) = r(c) exp2(2 F2).  A row of 8 voxels
// is walked in one piece (tier 0) or re-anchored after 4 voxels (tier 1).  The recurrence is safe unless the Gaussian is so
// thin along z that `steps` steps could climb from an underflowed start (exponent < -126) back above the alpha cut-off, or it
// has no finite culling box: such entries are evaluated exactly.  Voxel-space Gaussians are small (sigma ~ 1-3 voxels): with
// 3 steps the bound is |F2| <= ~5 (sigma_z >= 0.37 voxel), with 7 steps |F2| <= ~0.95 (sigma_z >= 0.87 voxel).
constexpr int VOX_RECUR_STEPS = 4;
__device__ __forceinline__ bool recurrence_safe3(float F2, float L, int steps)
{
    // head-room sqrt(126 + L), not sqrt(126): the exponent that underflows includes L (see row_tier in raster_state.hpp)
    const float smax = (__builtin_amdgcn_sqrtf(fmaxf(125.5f + fminf(L, 0.f), 0.f)) - __builtin_amdgcn_sqrtf(fmaxf(L - LOG2_ALPHA_MIN_3D, 0.f) + 1.0f)) *
                       (1.0f / (float)steps);
    return smax > 0.f && fabsf(F2) <= smax * smax;
}
__device__ __forceinline__ bool needs_exact_row3(float F2, float L, float hx)
{
    return !recurrence_safe3(F2, L, VOX_RECUR_STEPS - 1) || !(hx < 3.0e38f);
}
// Round 4: the lane-per-entry step also walks the slab's ROWS by recurrence (rows 4..7 up from row 4, rows 3..0 down from row
// 3: VOX_RECUR_YSTEPS steps), then each row along z as before.  sqrt(-exponent) is a norm of the offset (positive definite
// conic), so a voxel reached from an anchor by ny row steps and nz voxel steps differs from it by at most
// ny sqrt|D2| + nz sqrt|F2| in that norm: an underflowed anchor (exponent < -126) cannot precede a voxel above the cut-off
// if that sum stays below sqrt(126 + L) - sqrt(L - log2(1e-6)).  At 256^3 / 300k Gaussians 0.3 % of the instances fail this
// (1.9 % at 128^3 / 50k, none of the trained clouds'); they are evaluated exactly like the thin-along-z ones.
constexpr int VOX_RECUR_YSTEPS = 3;
__device__ __forceinline__ bool needs_exact_slab3(float D2, float F2, float L, float hx)
{
    const float smax = __builtin_amdgcn_sqrtf(fmaxf(125.5f + fminf(L, 0.f), 0.f)) - __builtin_amdgcn_sqrtf(fmaxf(L - LOG2_ALPHA_MIN_3D, 0.f) + 1.0f);
    const float need = (float)VOX_RECUR_YSTEPS * __builtin_amdgcn_sqrtf(fabsf(D2)) + (float)(VOX_RECUR_STEPS - 1) * __builtin_amdgcn_sqrtf(fabsf(F2));
    return !(smax > 0.f && need <= smax) || !(hx < 3.0e38f);
}

// Workspace of the stick-first binning chain (voxel_sticks.hip), part of the geometry state: 18 bytes per Gaussian + 150 KB.
constexpr uint32_t VS_MAX_LISTS = 4096;     // lists (sticks of 2^shift consecutive tiles) per call: one LDS histogram
constexpr uint32_t VS_PRODUCER = 1024;      // Gaussians per producer workgroup (512: the same scatter time, twice the rows to scan: +6 us)
// Producer workgroups of the chain (cull + count, scatter): VS_PRODUCER threads, ONE workgroup per CU as long as a thread then
// owns at most VS_PER_THREAD_MAX Gaussians (293 workgroups of 1024 Gaussians on 256 CUs: the 37 CUs with two of them finished
// 17 us after the others); thread t of workgroup w owns Gaussians w * per_wg + it * VS_PRODUCER + t, it < ni, below (w + 1) * per_wg
constexpr uint32_t VS_PER_THREAD_MAX = 2;
constexpr uint32_t VS_BIG_GAUSSIAN = 256;   // tiles beyond which a Gaussian's tile cube is walked by its whole wave, a row per lane
struct VSGrid { uint32_t wgs, per_wg, ni; };
inline VSGrid vs_grid(int P, int cus)
{
    VSGrid g;
    const uint32_t n = (uint32_t)(P > 0 ? P : 1), c = (uint32_t)(cus > 0 ? cus : 1);
    const uint32_t one = (n + VS_PRODUCER - 1) / VS_PRODUCER;   // workgroups at one Gaussian per thread
    if (one <= c) { g.wgs = one; g.per_wg = VS_PRODUCER; g.ni = 1; return g; }
    g.per_wg = (n + c - 1) / c;
    if (g.per_wg > VS_PRODUCER * VS_PER_THREAD_MAX) g.per_wg = VS_PRODUCER * VS_PER_THREAD_MAX;
    g.ni = (g.per_wg + VS_PRODUCER - 1) / VS_PRODUCER;
    g.wgs = (n + g.per_wg - 1) / g.per_wg;   // <= ceil(P / VS_PRODUCER): the rows VoxelSticks reserves
    return g;
}

// the chain's device counters, bumped by many workgroups: a persistent block per (host thread, device, stream), zero between
// calls (the scan kernel's last workgroup resets them); shared with the small-grid path's counters (voxel_small_counter_block)
struct VSCounters {
    unsigned long long total;   // visible Gaussians << 40 | instances
    uint32_t maxlist;           // longest list
    uint32_t scan_done;         // scan workgroups that have finished
};
struct VoxelSticks {
    uint32_t *ctr;       // [16]  {big lists, short lists} for the sort kernel
    uint32_t *totals;    // [VS_MAX_LISTS] instances per list
    uint32_t *wgtot;     // [NW]  instances per producer workgroup
    uint4 *big, *small;  // the sort kernel's work lists {list, part, first instance, instances}: [bigcap] long lists (one entry per
                         // part of a list beyond one workgroup's capacity), [VS_MAX_LISTS] short ones
    size_t bigcap;
    uint32_t *H;         // [NW][stride] instances per (producer workgroup, list) -> after the scan: exclusive prefix over the workgroups
    size_t NW;
    size_t bytes;
    static VoxelSticks carve(char *chunk, int P)
    {
        VoxelSticks t;
        Bump b(chunk);
        t.NW = ((size_t)(P > 0 ? P : 1) + VS_PRODUCER - 1) / VS_PRODUCER;
        t.ctr = b.take<uint32_t>(16);
        t.totals = b.take<uint32_t>(VS_MAX_LISTS);
        t.wgtot = b.take<uint32_t>(t.NW);
        t.bigcap = VS_MAX_LISTS + (size_t)(P > 0 ? P : 0) / 8;
        t.big = b.take<uint4>(t.bigcap);
        t.small = b.take<uint4>(VS_MAX_LISTS);
        t.H = b.take<uint32_t>(t.NW * VS_MAX_LISTS);
        t.bytes = b.total();
        return t;
    }
};

struct VoxelGeom {
    float4 *rec;              // [3P] {x,y,z (voxel units), opacity} {a2,b2,c2,d2} {e2,f2,L,kz}: inverse covariance
                              //      (xx,xy,xz,yy,yz,zz) pre-scaled by -log2e/2 (diagonal) or -log2e (off-diagonal),
                              //      L = log2(opacity)
    float4 *ext;              // [P]  {hx,hyc,hzc,ky}: x half-extent (voxels) of the bounding box of alpha >= 1e-6; half-extents
                              //      of its central y-z cross-section; with kz: how the cross-section's centre moves with x
    uint32_t *depth_key;      // [P]  bits of world z, the arbitrary low sort word of the reference (Q10); 
    uint32_t *iota;           // [P]
    uint32_t *depth_sorted;   // [P]
    uint32_t *order;          // [P]  Gaussian ids in (z bits, id) order
    uint32_t *first;          // [P]  first instance of the Gaussian in the emission list
    uint4 *cube;              // [P]  {first, lo.x | lo.y << 16, lo.z | nx << 16, ny}: the Gaussian's tile cube (origin + x/y
                              //      extents in tiles), from which the backward recomputes an instance's emission index
    float *cov3D;             // [6P]
    uint32_t *tiles_touched;  // [P]
    uint32_t *host_words;     // [DW_COUNT] the words the host reads back: the control block at the start of dorder_temp
    uint32_t *offsets;        // [P]  inclusive scan of tiles_touched[order[j]]
    char *scan_temp;
    size_t scan_bytes;
    char *dorder_temp;        // depth order (bucket sort) workspace; starts with the control block = host_words
    size_t dorder_bytes;
    char *psort_temp;
    size_t psort_bytes;
    char *stick_temp;         // VoxelSticks
    size_t stick_bytes;
    size_t bytes;
    static VoxelGeom carve(char *chunk, int P)
    {
        VoxelGeom g;
        Bump b(chunk);
        g.rec = b.take<float4>(3 * (size_t)P);
        g.ext = b.take<float4>(P);
        g.depth_key = b.take<uint32_t>(P);
        g.iota = b.take<uint32_t>(P);
        g.depth_sorted = b.take<uint32_t>(P);
        g.order = b.take<uint32_t>(P);
        g.first = b.take<uint32_t>(P);
        g.cube = b.take<uint4>(P);
        g.cov3D = b.take<float>(6 * (size_t)P);
        g.tiles_touched = b.take<uint32_t>(P);
        g.offsets = b.take<uint32_t>(P);
        g.scan_bytes = scan_gather_temp_bytes(P);
        g.scan_temp = b.take<char>(g.scan_bytes);
        g.dorder_bytes = depth_order_temp_bytes((size_t)P);
        g.dorder_temp = b.take<char>(g.dorder_bytes);
        g.host_words = chunk ? depth_order_words(g.dorder_temp, (size_t)P) : nullptr;
        g.psort_bytes = sort_temp_bytes((size_t)P);   // radix fallback of the depth order
        g.psort_temp = b.take<char>(g.psort_bytes);
        g.stick_bytes = VoxelSticks::carve(nullptr, P).bytes;
        g.stick_temp = b.take<char>(g.stick_bytes);
        g.bytes = b.total();
        return g;
    }
};

struct VoxelBinning {
    uint32_t *tiles_unsorted, *tiles;   // [R]
    uint32_t *vals_unsorted;            // [R] Gaussian id per instance (emission order)
    uint32_t *inv;                      // [R] sorted position of emission index u; only written by the single-pass tile sort
                                        //     (<= 4096 tiles), for introspection -- nothing in the pipeline reads it
    uint32_t *point_list;               // [R]
    float *part;                        // [R*VPART_STRIDE] backward scratch
    char *sort_temp;
    size_t sort_bytes;
    size_t bytes;
    static VoxelBinning carve(char *chunk, size_t R)
    {
        VoxelBinning s;
        Bump b(chunk);
        s.tiles_unsorted = b.take<uint32_t>(R);
        s.tiles = b.take<uint32_t>(R);
        s.vals_unsorted = b.take<uint32_t>(R);
        s.inv = b.take<uint32_t>(R);
        s.point_list = b.take<uint32_t>(R);
        s.part = b.take<float>(R * VPART_STRIDE);
        s.sort_bytes = sort_temp_bytes(R);
        s.sort_temp = b.take<char>(s.sort_bytes);
        s.bytes = b.total();
        return s;
    }
};

struct VoxelImage {
    uint2 *ranges;          // [T3]
    uint32_t *chunk_base;   // [T3+1]
    uint4 *work_tile;       // [NW]
    float *partial;         // [NW*512]
    uint32_t *partial_last; // [NW*512] debug only
    uint32_t *n_contrib;    // [V], debug only
    char *work_temp;       // scratch of the parallel work-list construction
    size_t NW;
    size_t R;               // instances (host-side copy: launch sizing)
    size_t bytes;
    static VoxelImage carve(char *chunk, size_t T, size_t V, size_t R, bool debug, uint32_t list_chunk /* vox_chunk_for(gy, gz) */)
    {
        VoxelImage s;
        Bump b(chunk);
        s.R = R;
        s.NW = R / list_chunk + T;
        s.ranges = b.take<uint2>(T);
        s.chunk_base = b.take<uint32_t>(T + 1);
        s.work_tile = b.take<uint4>(s.NW);
        s.partial = b.take<float>(s.NW * 512);
        s.partial_last = b.take<uint32_t>(debug ? s.NW * 512 : 0);
        s.n_contrib = b.take<uint32_t>(debug ? V : 0);
        s.work_temp = b.take<char>(build_work_temp_bytes(T));
        s.bytes = b.total();
        return s;
    }
};

// The grid of a call.  An ordinary call: the volume of the settings.  An x-SLAB call (r2_voxel_forward_slab: tile layers [tile_x0,
// tile_x1) of the full grid, the unit of the sharded query): (nx, gx) describe the slab -- the output block is [nx, ny, nz], the
// tile grid gx x gy x gz, tile and voxel indices count from the slab's first layer -- while (sx, cx, fnx, fgx) stay the FULL
// volume's, so that every float the preprocess and the render kernels form (voxel size, voxel-space position, radii, tile cube,
// distance to a voxel centre = position - (ox + slab-local index)) is the unsharded call's, bit for bit.
struct VoxelGrid {
    int nx, ny, nz;
    float sx, sy, sz;
    float cx, cy, cz;
    int gx, gy, gz;
    int ox;         // voxel x index (full grid) of the slab's first voxel: tile_x0 * TILE3D; 0 for an ordinary call
    int fnx, fgx;   // nVoxel_x and tiles along x of the full grid (= nx, gx for an ordinary call)
    bool is_slab() const { return ox != 0 || fgx != gx; }
};

int launch_voxel_preprocess(const VoxelGeom &g, const VoxelGrid &v, int P, const float *means3D, const float *scales,
                            float scale_modifier, const float *rotations, const float *opacities,
                            const float *cov3D_precomp, int *radii_x, int *radii_y, int *radii_z, const DepthReg &reg,
                            bool store_cov3D, hipStream_t s);
// small grids (<= 64 tiles): preprocess + survivor list (voxel_geom.hip), per-tile lists from it (voxel_small.hip)
int launch_voxel_preprocess_small(const VoxelGeom &g, const VoxelGrid &v, int P, const float *means3D, const float *scales,
                                  float scale_modifier, const float *rotations, const float *opacities,
                                  const float *cov3D_precomp, int *radii_x, int *radii_y, int *radii_z, uint4 *surv,
                                  unsigned long long *counter, uint32_t *mailbox, uint32_t seq, hipStream_t s);
constexpr size_t VOX_SMALL_MAX_TILES = 64;         // 4 x 4 x 4 tiles: the 32^3 TV patch
// For such grids the FORWARD leaves the per-instance tile ids (binning state `tiles`) behind, whichever path it took and
// whatever its debug flag; the backward of larger single-pass grids fills them from the ranges itself.  A static rule of the
// grid alone: the backward cannot ask which forward path ran, nor with which flags.
inline bool voxel_forward_fills_tiles(size_t T) { return T <= VOX_SMALL_MAX_TILES; }
constexpr uint32_t VOX_SMALL_MAX_SURVIVORS = 8192;  // a tile's list (<= all survivors) is sorted in LDS: 2 x 8 bytes per entry
// -> num_rendered (>= 0), a negative R2_ERR_* code, or VOX_SMALL_NOT_TAKEN: the caller runs the general pipeline
constexpr int VOX_SMALL_NOT_TAKEN = -1000;
constexpr uint32_t VOX_SMALL_MARK = 0x5A11u;      // DW_USER word of a state produced by the small-grid path (introspection / tests)
void voxel_small_release();   // the calling thread's counter blocks of the small-grid path (r2_thread_release)
int voxel_forward_small(r2_alloc_fn binningBuffer, void *binning_user, r2_alloc_fn imageBuffer, void *image_user,
                        const VoxelGeom &geom, const VoxelGrid &v, int P, const float *means3D, const float *opacities,
                        const float *scales, float scale_modifier, const float *rotations, const float *cov3D_precomp,
                        float *out_volume, int *radii_x, int *radii_y, int *radii_z, hipStream_t s);
// large grids (more than 4096 tiles, e.g. the 256^3 query): stick-first binning (voxel_sticks.hip).
// -> num_rendered (>= 0), a negative R2_ERR_* code, VOX_STICKS_NOT_TAKEN (nothing was launched: run the general pipeline) or
// VOX_STICKS_FALLBACK (the preprocess has run, without depth registration: continue with the general pipeline's un-hinted branch)
constexpr int VOX_STICKS_NOT_TAKEN = -1001, VOX_STICKS_FALLBACK = -1002;
constexpr uint32_t VOX_STICKS_MARK = 0x571Cu;   // DW_USER word of a state produced by that chain (introspection / tests)
void voxel_sticks_release();   // the calling thread's notes about that chain (r2_thread_release)
// 64 bytes of device memory, zero between calls, owned by the calling thread for (current device, stream); nullptr: none to be had
unsigned long long *voxel_small_counter_block(int dev, hipStream_t s);
void voxel_counter_block_clean(int dev, hipStream_t s);   // ... and its kernels have put the zeros back (else the next call zero-fills it)
// switches and device requirements of the two fast chains, for voxel_forward_choice (dispatch.hpp)
bool voxel_small_switched_on();
bool voxel_small_lds_ok();
bool voxel_sticks_switched_on();
bool voxel_sticks_lds_ok();
// the preprocess as two kernels: (1) everything the binning needs + the per-(workgroup, list) instance counts of the stick-first
// chain (lists = tile id >> shift; H[workgroup][stride], wgtot[workgroup], the call's totals into ctr), (2) the render records,
// in one launch with the column scan of H (whose last workgroup posts the totals to the host mailbox)
int launch_voxel_cull_count(const VoxelGeom &g, const VoxelGrid &v, int P, const VSGrid &grid, const float *means3D, const float *scales,
                            float scale_modifier, const float *rotations, const float *cov3D_precomp, int *radii_x, int *radii_y,
                            int *radii_z, uint32_t shift, uint32_t stride, uint32_t *H, uint32_t *wgtot, VSCounters *ctr, hipStream_t s);
int launch_voxel_scan_records(const VoxelGeom &g, const VoxelGrid &v, int P, const VSGrid &grid, const float *means3D,
                              const float *opacities, const float *cov3D_precomp, uint32_t *H, uint32_t rows, uint32_t stride,
                              uint32_t *totals, VSCounters *ctr, uint32_t *mailbox, uint32_t seq, hipStream_t s);
int voxel_forward_sticks(r2_alloc_fn binningBuffer, void *binning_user, r2_alloc_fn imageBuffer, void *image_user,
                         const VoxelGeom &geom, const VoxelGrid &v, int P, const float *means3D, const float *opacities,
                         const float *scales, float scale_modifier, const float *rotations, const float *cov3D_precomp,
                         float *out_volume, int *radii_x, int *radii_y, int *radii_z, uint32_t stick_shift, hipStream_t s);
int launch_voxel_duplicate(const VoxelGeom &g, const VoxelBinning &b, const VoxelGrid &v, int P, const int *radii_x,
                           const int *radii_y, const int *radii_z, const uint32_t *nvis, hipStream_t s, uint2 *zero_ranges = nullptr, size_t zero_T = 0);
// the seven gradient arrays of the voxelizer backward, for a kernel that zero-fills them
struct ZeroArrays {
    float *p[7];
    size_t n[7];   // floats
};
int launch_voxel_geom_backward(const VoxelGeom &g, const VoxelGrid &v, int P, const int *radii_x, const int *radii_y,
                               const int *radii_z, const float *cov3D, const float *scales, const float *rotations,
                               float scale_modifier, const float *part, float *dL_dconic3D, float *dL_dmean3D_norm,
                               float *dL_dopacity, float *dL_dmean3D, float *dL_dcov3D, float *dL_dscale, float *dL_drot,
                               hipStream_t s, bool rows_already_zero = false);
// tile lists shorter than this get no forward work item (a light kernel renders them); 0 in debug mode
uint32_t voxel_short_list_min(bool debug);
// Optional side job of the forward's last kernel (small-grid path): move the lists from where they were built into the binning /
// image state (n == 0: nothing to do).
struct VoxelPublish {
    const uint32_t *src_plist, *src_tiles, *src_chunk_base;
    const uint2 *src_ranges;
    const uint4 *src_work;
    uint32_t *dst_plist, *dst_tiles, *dst_chunk_base;
    uint2 *dst_ranges;
    uint4 *dst_work;
    uint32_t R, T, NW, n;   // n = max(R, T + 1, NW)
};
int launch_voxel_render_forward(const VoxelGeom &g, const VoxelBinning &b, const VoxelImage &im, const VoxelGrid &v,
                                float *out_volume, bool write_ncontrib, hipStream_t s, bool no_short_kernel = false,
                                const VoxelPublish *publish = nullptr);
// zero (optional): gradient arrays to zero-fill with EXTRA workgroups of the same launch (patches: the render backward of a
// 32^3 patch leaves most of the machine idle, and a zero-fill launch of its own was 10 of the TV step's 91 us)
int launch_voxel_render_backward(const VoxelGeom &g, const VoxelBinning &b, const VoxelGrid &v, size_t R,
                                 const float *dL_dvol, hipStream_t s, const ZeroArrays *zero = nullptr);


}  // namespace r2
