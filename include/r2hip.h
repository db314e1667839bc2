/*
 * r2hip.h -- C ABI of libr2hip.so, the MI355X (gfx950) implementation of the R2-Gaussian hot path:
 * differentiable X-ray rasterizer, 3D voxelizer and simple-knn.
 *
 * Every entry point replaces one function of the reference's native layer L0/L1
 * (paths relative to r2_gaussian/submodules/xray-gaussian-rasterization-voxelization/ = SUB):
 *
 *   r2_raster_forward     <- CudaRasterizer::Rasterizer::forward      SUB/cuda_rasterizer/rasterizer.h:36-56,  rasterizer_impl.cu:196-331
 *   r2_raster_backward    <- CudaRasterizer::Rasterizer::backward     SUB/cuda_rasterizer/rasterizer.h:58-85,  rasterizer_impl.cu:335-421
 *   r2_mark_visible       <- CudaRasterizer::Rasterizer::markVisible  SUB/cuda_rasterizer/rasterizer.h:29-34,  rasterizer_impl.cu:141-153
 *   r2_voxel_forward      <- CudaVoxelizer::Voxelizer::forward        SUB/cuda_voxelizer/voxelizer.h:28-47,    voxelizer_impl.cu:171-302
 *   r2_voxel_backward     <- CudaVoxelizer::Voxelizer::backward       SUB/cuda_voxelizer/voxelizer.h:49-72,    voxelizer_impl.cu:307-389
 *   r2_knn_dist2          <- simple_knn._C.distCUDA2 (un-vendored submodule; call site r2_gaussian/gaussian/gaussian_model.py:145-150)
 *
 * Conventions (identical to the reference's L0):
 *   - all data pointers are DEVICE pointers to contiguous float32 / int32 arrays owned by the caller;
 *   - "absent" optional inputs (scales/rotations vs cov3D_precomp) are passed as NULL;
 *   - 4x4 matrices are 16 floats indexed m[col*4+row] (== row-major memory of the transposed
 *     matrices torch hands over: world_view_transform, full_proj_transform);
 *   - the library never allocates result/state memory: it asks the caller for bytes through the
 *     three r2_alloc_fn callbacks, the C form of the reference's std::function<char*(size_t)>
 *     (SUB/utility.h:7-13).  The layout inside those buffers is private to the library;
 *   - gradient outputs of the backward calls are FULLY WRITTEN by the library (all-zero rows for culled
 *     Gaussians, for the scale/rotation gradients on the cov3D_precomp path, and the unused third component of
 *     dL_dmean2D): the zero-initialisation the reference's torch boundary performs
 *     (SUB/rasterize_points.cu:124-131, SUB/voxelize_points.cu:130-136) is not required;
 *   - `stream` is a hipStream_t (NULL = the null stream).  Work is enqueued on it; the forward
 *     calls synchronise that stream once to learn num_rendered (the reference's cudaMemcpy D2H,
 *     rasterizer_impl.cu:279), the backward calls do not synchronise;
 *   - return value: >= 0 on success (forward: num_rendered), < 0 = -(hipError_t) or R2_ERR_*;
 *     r2_last_error() gives a message.  With debug != 0 the stream is synchronised and checked after
 *     every stage (the reference's CHECK_CUDA, SUB/cuda_rasterizer/auxiliary.h:170-177).
 */
#ifndef R2HIP_H
#define R2HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define R2_API __attribute__((visibility("default")))
#else
#define R2_API
#endif

#define R2_ABI_VERSION 3
#define R2_ERR_INVALID (-10001) /* bad argument (NULL where data is required, negative size ...) */
#define R2_ERR_ALLOC   (-10002) /* an r2_alloc_fn callback returned NULL */

/* Returns a device pointer to at least `bytes` bytes (128-byte aligned), valid until the matching
 * backward call has been enqueued.  `user` is passed through untouched.
 * The rasterizer forward may call the binning and image callbacks a SECOND time within one call (tile-first chain: the state
 * was sized by a prediction that fell short and is asked for again with the exact size, see r2_tile_first_control).  Kernels
 * enqueued on `stream` before the second request still hold pointers into the first buffer (they find the true count on the
 * device and do nothing, but they do read and write a few words of it), so a callback that releases or reuses the first buffer
 * must do so IN STREAM ORDER on `stream` -- true of a stream-ordered allocator such as torch's caching allocator on the
 * current stream, of hipFreeAsync on `stream`, and of a synchronous hipFree; a callback that recycles memory on ANOTHER
 * stream must keep the first buffer alive until `stream` has passed the forward.  The buffer the backward must be given is the
 * one returned LAST. */
typedef char *(*r2_alloc_fn)(size_t bytes, void *user);

R2_API int r2_abi_version(void);
R2_API const char *r2_last_error(void);

/* Alignment.  All pointers are device pointers to float / int arrays and must be 4-byte aligned; the kernels move the
 * following arrays 16 bytes at a time, so THESE must be 16-byte aligned (any fresh torch / hipMalloc allocation is):
 *   rotations [P,4], dL_dpix [H,W] (rasterizer backward, when width % 16 == 0), dL_dconic [P,2,2], dL_drot [P,4] (both
 *   backwards), and the state buffers handed out by the r2_alloc_fn callbacks (128-byte aligned).
 * The torch boundaries (r2_gaussian_amd/_C.py, csrc/torch_shim.cpp) copy an input whose data pointer is not 16-byte
 * aligned (e.g. a view into a flat parameter buffer at an odd offset) before passing it down.
 * Devices / threads.  Calls take the caller's HIP stream; the device that stream belongs to must be current (hipSetDevice)
 * in the calling thread, as for every HIP API that takes a stream.  A host thread may drive several devices.
 * num_rendered.  The forward calls return the number of (tile, Gaussian) instances as a non-negative int; a scene whose
 * instance count does not fit 31 bits is rejected with R2_ERR_INVALID (the reference's int num_rendered has the same range).
 */

/* ---- rasterizer ------------------------------------------------------------------------------ */
R2_API int r2_raster_forward(
    r2_alloc_fn geometryBuffer, void *geometry_user,
    r2_alloc_fn binningBuffer, void *binning_user,
    r2_alloc_fn imageBuffer, void *image_user,
    int P, int width, int height,
    const float *means3D,      /* [P,3] */
    const float *opacities,    /* [P]   activated density */
    const float *scales,       /* [P,3] or NULL */
    float scale_modifier,
    const float *rotations,    /* [P,4] (r,x,y,z) or NULL */
    const float *cov3D_precomp,/* [P,6] or NULL */
    const float *viewmatrix,   /* [16] */
    const float *projmatrix,   /* [16] */
    const float *cam_pos,      /* [3], unused by the X-ray path (kept for signature parity) */
    float tan_fovx, float tan_fovy,
    int prefiltered,
    int mode,                  /* 0 parallel beam, 1 cone beam */
    float *out_color,          /* [1,H,W] */
    int *radii,                /* [P] */
    int debug,
    void *stream);

R2_API int r2_raster_backward(
    int P, int R, int width, int height,
    const float *means3D, const float *scales, float scale_modifier, const float *rotations,
    const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix, const float *campos,
    float tan_fovx, float tan_fovy,
    const int *radii,
    char *geom_buffer, char *binning_buffer, char *img_buffer,
    const float *dL_dpix,      /* [1,H,W] */
    float *dL_dmean2D,         /* [P,3] (z stays 0) */
    float *dL_dconic,          /* [P,2,2] (slots 0,1,3); 16-byte aligned */
    float *dL_dopacity,        /* [P,1] */
    float *dL_dmu,             /* [P,1] */
    float *dL_dmean3D,         /* [P,3] */
    float *dL_dcov3D,          /* [P,6] */
    float *dL_dscale,          /* [P,3] */
    float *dL_drot,            /* [P,4] */
    int mode, int debug, void *stream);

R2_API int r2_mark_visible(int P, const float *means3D, const float *viewmatrix, const float *projmatrix,
                    uint8_t *present /* [P] bool */, void *stream);

/* ---- rasterizer, batched views (NEW functionality: the reference renders one view per call) ----
 * V views of the SAME Gaussians (same detector size, tan_fov and mode; V camera poses) through one pass of the pipeline:
 * the views become V * P "view instances" on a tile grid that stacks the views' grids, so every latency-bound stage
 * (preprocess, depth order, instance emission, tile sort) runs once on V times the work.  A trainer that accumulates
 * several views per optimiser step (view-sharded data parallelism) calls these instead of V single-view
 * calls.  Every view is evaluated with exactly the arithmetic of r2_raster_forward / r2_raster_backward: out_color[v] and
 * radii[v] are bit-identical to the single-view call's, tile lists too.
 *   forward : viewmatrices / projmatrices [V,16]; out_color [V,H,W]; radii [V,P]; returns num_rendered over all views;
 *             width * height need not be tile-aligned.
 *   backward: dL_dpix [V,H,W]; per view: dL_dmean2D [V,P,3], dL_dconic [V,P,2,2], dL_dmu [V,P]; SUMMED over the views (in
 *             view order, deterministic): dL_dopacity [P], dL_dmean3D [P,3], dL_dcov3D [P,6], dL_dscale [P,3], dL_drot [P,4]. */
R2_API int r2_raster_forward_batch(
    r2_alloc_fn geometryBuffer, void *geometry_user,
    r2_alloc_fn binningBuffer, void *binning_user,
    r2_alloc_fn imageBuffer, void *image_user,
    int P, int V, int width, int height,
    const float *means3D, const float *opacities, const float *scales, float scale_modifier, const float *rotations,
    const float *cov3D_precomp,
    const float *viewmatrices, /* [V,16] */
    const float *projmatrices, /* [V,16] */
    float tan_fovx, float tan_fovy, int mode,
    float *out_color,          /* [V,H,W] */
    int *radii,                /* [V,P] */
    int debug, void *stream);

R2_API int r2_raster_backward_batch(
    int P, int V, int R, int width, int height,
    const float *means3D, const float *scales, float scale_modifier, const float *rotations,
    const float *cov3D_precomp, const float *viewmatrices, const float *projmatrices,
    float tan_fovx, float tan_fovy,
    const int *radii,          /* [V,P] */
    char *geom_buffer, char *binning_buffer, char *img_buffer,
    const float *dL_dpix,      /* [V,H,W] */
    float *dL_dmean2D,         /* [V,P,3] */
    float *dL_dconic,          /* [V,P,2,2]; 16-byte aligned */
    float *dL_dopacity,        /* [P,1]   summed over the views */
    float *dL_dmu,             /* [V,P] */
    float *dL_dmean3D,         /* [P,3]   summed */
    float *dL_dcov3D,          /* [P,6]   summed */
    float *dL_dscale,          /* [P,3]   summed */
    float *dL_drot,            /* [P,4]   summed */
    int mode, int debug, void *stream);

/* ---- loss stack of the training iteration (r2_gaussian/utils/loss_utils.py:19-104, train.py:118-147) ----
 * r2_loss_l1_ssim: loss = w_l1 * mean|img - gt| + w_ssim * (1 - SSIM(img, gt)) (11x11 Gaussian window, sigma 1.5, zero
 * padding) of one [height,width] projection AND its gradient dL/dimg, in two launches; scalars = {l1 mean, ssim mean, loss}.
 * r2_loss_tv3d: tv = tv_3d_loss(vol, "mean") of a [nx,ny,nz] volume and dL/dvol = weight * d tv / d vol; scalars = {tv,
 * weight * tv}.  scratch: device floats, at least r2_loss_*_scratch_floats(...).  Sums are formed in a fixed order. */
R2_API size_t r2_loss_l1_ssim_scratch_floats(int width, int height);
R2_API int r2_loss_l1_ssim(int width, int height, const float *img, const float *gt, float w_l1, float w_ssim,
                           float *dL_dimg, float *scratch, float *scalars /* [3] */, void *stream);
R2_API size_t r2_loss_tv3d_scratch_floats(int nx, int ny, int nz);
R2_API int r2_loss_tv3d(int nx, int ny, int nz, const float *vol, float weight, float *dL_dvol, float *scratch,
                        float *scalars /* [2] */, void *stream);

/* ---- adaptive density control on the device (r2_gaussian/gaussian/gaussian_model.py:320-556, train.py:151-168) ----
 * r2_densify_stats: max_radii2D / xyz_gradient_accum / denom update of one rendered view (in place, one launch).
 * r2_densify_classify + r2_densify_emit: densify_and_prune -- clone (small Gaussians with a large view-space gradient; both
 * copies get half the density), split (large ones: two children sampled from N(0, scale) in the local frame, scale / 1.6,
 * half the density, parent removed), prune (density below density_min, outside the box) -- with the Adam moments carried
 * along (zeros for new rows) and the statistics reset, written in the reference's row order.  classify decides, counts and
 * synchronises the stream ONCE to return the four survivor counts (originals, clones, first children, second children);
 * the caller allocates sum(counts) rows and calls emit with the same arguments and the same scratch.  normals: [2,P,3]
 * N(0,1) samples indexed by the PARENT's row (only rows of split parents are read).  scale_lo < scale_hi: bounded-sigmoid
 * scaling activation, else exp.  max_screen_size / max_scale: the reference's optional prune thresholds (rows whose
 * max_radii2D / largest activated scale exceed them are pruned, gaussian_model.py:540-545); <= 0 switches them off (None).
 * params / exp_avg / exp_avg_sq (+ _out): 4 device pointers each in the order xyz[.,3], density[.,1], scaling[.,3], rotation[.,4]. */
R2_API int r2_densify_stats(int P, const int *radii, const float *dL_dmeans2D /* [P,3] */, float *max_radii2D, float *grad_accum,
                            float *denom, void *stream);
R2_API size_t r2_densify_scratch_bytes(int P);
R2_API int r2_densify_classify(int P, const float *xyz, const float *density, const float *scaling, const float *rotation,
                               const float *max_radii2D, const float *grad_accum, const float *denom, const float *normals,
                               float grad_thr, float scale_thr, float density_min,
                               const float *bbox_host /* 6 host floats: lo xyz, hi xyz */, float scale_lo, float scale_hi,
                               int do_densify, float max_screen_size, float max_scale, void *scratch,
                               unsigned int *counts_host /* [4] */, void *stream);
R2_API int r2_densify_emit(int P, const float *const *params, const float *const *exp_avg, const float *const *exp_avg_sq,
                           const float *max_radii2D, const float *grad_accum, const float *denom, const float *normals,
                           float grad_thr, float scale_thr, float density_min, const float *bbox_host, float scale_lo,
                           float scale_hi, int do_densify, float max_screen_size, float max_scale, const void *scratch,
                           float *const *params_out, float *const *exp_avg_out,
                           float *const *exp_avg_sq_out, float *max_radii2D_out, float *grad_accum_out, float *denom_out,
                           void *stream);

/* ---- voxelizer ------------------------------------------------------------------------------- */
R2_API int r2_voxel_forward(
    r2_alloc_fn geometryBuffer, void *geometry_user,
    r2_alloc_fn binningBuffer, void *binning_user,
    r2_alloc_fn imageBuffer, void *image_user,
    int P,
    int nVoxel_x, int nVoxel_y, int nVoxel_z,
    float sVoxel_x, float sVoxel_y, float sVoxel_z,
    float center_x, float center_y, float center_z,
    const float *means3D, const float *opacities, const float *scales, float scale_modifier,
    const float *rotations, const float *cov3D_precomp,
    int prefiltered,
    float *out_volume,         /* [nx,ny,nz] */
    int *radii_x, int *radii_y, int *radii_z, /* [P] each */
    int debug, void *stream);

R2_API int r2_voxel_backward(
    int P, int R,
    int nVoxel_x, int nVoxel_y, int nVoxel_z,
    float sVoxel_x, float sVoxel_y, float sVoxel_z,
    float center_x, float center_y, float center_z,
    const float *means3D, const float *scales, float scale_modifier, const float *rotations,
    const float *cov3D_precomp,
    const int *radii_x, const int *radii_y, const int *radii_z,
    char *geom_buffer, char *binning_buffer, char *img_buffer,
    const float *dL_dvol,      /* [nx,ny,nz] */
    float *dL_dmean3D_norm,    /* [P,3] */
    float *dL_dconic3D,        /* [P,6] */
    float *dL_dopacity,        /* [P,1] */
    float *dL_dmean3D,         /* [P,3] */
    float *dL_dcov3D,          /* [P,6] */
    float *dL_dscale,          /* [P,3] */
    float *dL_drot,            /* [P,4] */
    int debug, void *stream);

/* ---- voxelizer, one x-slab of the grid (NEW functionality: the unit of the sharded full-volume query) ----
 * Tile layers [tile_x0, tile_x1) along x (layers of 8 voxels; 0 <= tile_x0 < tile_x1 <= ceil(nVoxel_x / 8)) of the volume the
 * other arguments describe -- nVoxel / sVoxel / center are the FULL volume's, exactly as for r2_voxel_forward.  The call evaluates
 * the full grid's arithmetic (voxel size, voxel-space positions, radii, tile cubes, distances to the voxel centres) and bins /
 * renders only the slab's tiles: out_volume is the [min(8 tile_x1, nVoxel_x) - 8 tile_x0, ny, nz] block of the full volume,
 * BIT-IDENTICAL to the same voxels of r2_voxel_forward's result, and the slab's tile lists are the full call's lists of those tiles
 * (reference: ONE grid with one arithmetic, test.py:105-112, SUB/cuda_voxelizer/forward.cu:58-178, voxelizer_impl.cu:54-101).
 * Slabs are independent: no exchange between them.  radii_{x,y,z}: the full call's radii for Gaussians with a tile in the slab, 0
 * for the others.  Returns the slab's num_rendered.  r2_voxel_forward(...) == r2_voxel_forward_slab(..., 0, ceil(nVoxel_x / 8), ...).
 * The backward takes the same two numbers; dL_dvol is the slab's block. */
R2_API int r2_voxel_forward_slab(
    r2_alloc_fn geometryBuffer, void *geometry_user,
    r2_alloc_fn binningBuffer, void *binning_user,
    r2_alloc_fn imageBuffer, void *image_user,
    int P,
    int nVoxel_x, int nVoxel_y, int nVoxel_z,
    float sVoxel_x, float sVoxel_y, float sVoxel_z,
    float center_x, float center_y, float center_z,
    int tile_x0, int tile_x1,
    const float *means3D, const float *opacities, const float *scales, float scale_modifier,
    const float *rotations, const float *cov3D_precomp,
    int prefiltered,
    float *out_volume,         /* [slab nx,ny,nz] */
    int *radii_x, int *radii_y, int *radii_z, /* [P] each */
    int debug, void *stream);

R2_API int r2_voxel_backward_slab(
    int P, int R,
    int nVoxel_x, int nVoxel_y, int nVoxel_z,
    float sVoxel_x, float sVoxel_y, float sVoxel_z,
    float center_x, float center_y, float center_z,
    int tile_x0, int tile_x1,
    const float *means3D, const float *scales, float scale_modifier, const float *rotations,
    const float *cov3D_precomp,
    const int *radii_x, const int *radii_y, const int *radii_z,
    char *geom_buffer, char *binning_buffer, char *img_buffer,
    const float *dL_dvol,      /* [slab nx,ny,nz] */
    float *dL_dmean3D_norm, float *dL_dconic3D, float *dL_dopacity, float *dL_dmean3D, float *dL_dcov3D, float *dL_dscale,
    float *dL_drot,
    int debug, void *stream);

/* ---- simple-knn ------------------------------------------------------------------------------ */
/* mean of the 3 smallest squared distances to the other points; out[P].  The exact uniform-grid search (P >= 4096) works inside
 * a caller-provided workspace of r2_knn_workspace_bytes(P) bytes (any alignment >= 256 B); without one (NULL / too small) the
 * exhaustive O(P^2) kernel runs, which needs none.  Synchronises the stream twice (bounding box, fullest cell): it is called
 * once per training run (gaussian_model.py:145-150). */
R2_API size_t r2_knn_workspace_bytes(int P);
R2_API int r2_knn_dist2_ws(int P, const float *points /* [P,3] */, float *out /* [P] */, void *workspace, size_t workspace_bytes,
                           void *stream);
/* the reference's signature: obtains the workspace itself (hipMalloc; hipFree after waiting for the stream) -- the one entry
 * point of the library that allocates; callers with an allocator use the two above */
R2_API int r2_knn_dist2(int P, const float *points /* [P,3] */, float *out /* [P] */, void *stream);

/* ---- measurement: per-stage HIP-event timing on the caller's stream ---------------------------- */
/* bit i of stage_mask enables stage i (0 = off, the default: no events are recorded).  Enabled stages are
 * bracketed by hipEventRecord on the stream they are launched on; r2_profile_read synchronises the recorded
 * events and returns accumulated milliseconds and launch counts per stage (arrays of r2_profile_stage_count()
 * entries).  Not thread-safe; meant for bench.py. */
R2_API void r2_profile_enable(unsigned long long stage_mask);
R2_API int r2_profile_stage_count(void);
R2_API const char *r2_profile_stage_name(int stage);
R2_API int r2_profile_read(double *total_ms, long long *counts, int reset);
/* host time spent busy-waiting at the forward passes' synchronisation point (the D2H read of num_rendered), and the
 * number of such waits: long waits = GPU-bound, short waits = the host is the bottleneck. */
R2_API int r2_sync_wait_stats(double *total_us, long long *calls, int reset);
/* host time the forward passes spent before that wait (launching the first kernels) and after it (allocation callbacks +
 * launching the rest), accumulated over `calls` forward passes */
R2_API int r2_profile_host(double *pre_sync_us, double *post_sync_us, long long *calls, int reset);

/* ---- FDK reconstruction for the initialisation --------------------------------------------------------
 * Replaces tigre.algorithms.fdk as called by recon_volume() (r2_gaussian/utils/ct_utils.py:17-27) from init_pcd()
 * (initialize_pcd.py:36-90); TIGRE is a third-party CUDA toolbox outside the reference tree.
 * r2_fdk_filter: cosine pre-weight (cone != 0: DSD / sqrt(DSD^2 + u^2 + v^2) at the pixel centres, pixel size du x dv) and
 *   ramp filter along detector rows: out[v][i] = scale * sum_j w(j,v) projs[v][j] * taps[i - j + W - 1], taps = the 2W-1
 *   spatial taps of the (windowed) ramp (host side: r2_gaussian_amd/fdk.py:ramp_taps), scale = (DSD/DSO)(2 pi/V)/(4 du).
 *   The result is stored transposed: filtered_t[V][W][H].
 * r2_fdk_backproject: vol[nx][ny][nz] = sum over the views (in order) of the bilinear sample (zero outside the detector) of
 *   filtered_t at the projection of the voxel centre, times (DSO / U)^2 for cone beams (U = p_hom.w, the depth along the
 *   central ray).  projmatrices [V,16] are the full_proj_transform matrices the rasterizer takes (same memory layout), voxel
 *   centres follow the voxelizer: center - sVoxel/2 + (i + 0.5) dVoxel. */
R2_API int r2_fdk_filter(int V, int H, int W, const float *projs /* [V,H,W] */, const float *taps /* [2W-1] */, float scale,
                         int cone, float DSD, float du, float dv, float *filtered_t /* [V,W,H] */, void *stream);
R2_API int r2_fdk_backproject(int V, int H, int W, const float *filtered_t, const float *projmatrices, int cone, float DSO,
                              int nx, int ny, int nz, float sVoxel_x, float sVoxel_y, float sVoxel_z, float center_x,
                              float center_y, float center_z, float *vol /* [nx,ny,nz] */, void *stream);

/* The forward passes order the Gaussians by depth with a bucket sort whose bucket boundaries follow the depth range seen
 * by the previous call with the same P (a per-thread hint: it saves five kernel launches and hides the num_rendered
 * read-back).  Results never depend on it -- both paths produce the exact (depth, id) order.  mode 0: never use hints,
 * 1: use them (default; the environment variable R2_DEPTH_HINT=0 also switches them off), 2: forget the history. */
R2_API void r2_depth_hint_control(int mode);

/* Tile-first binning of the rasterizer forward (csrc/raster_tilefirst.hip): single-view calls whose instance count the
 * calling thread can predict from its recent calls with the same P and detector size skip the global depth order -- instances
 * are counted and scattered per tile and every tile list is sorted on (depth, id) on its own -- and size the binning / image
 * state by that prediction (the exact count is still returned; a prediction that falls short only costs a second pass).
 * point_list, ranges, images and gradients are identical on both chains.  mode 0: never, 1: when applicable (default; the
 * environment variable R2_TILE_FIRST=0 also switches it off), 2: forget the calling thread's predictions (its next call of any
 * size takes the general chain). */
R2_API void r2_tile_first_control(int mode);
/* Deferred num_rendered (NEW; opt-in; SURVEY.md section 7 "kill the D2H sync", RAS/rasterizer_impl.cu:279).  mode 1 (or the
 * environment variable R2_DEFER_COUNT=1): a rasterizer forward on the tile-first chain returns WITHOUT waiting for the device --
 * the state is sized by the prediction + 50 %, every kernel is enqueued, and the value returned in place of num_rendered is a
 * TOKEN (>= 0x40000000) that the matching r2_raster_backward / _backward_batch accepts as its R: the backward reads the true count,
 * which the forward's second kernel posted to pinned host memory long before (any host thread may call it).  Images, state and
 * gradients are those of the waiting mode.  The price: a prediction that falls short cannot be repaired by a second pass any
 * more -- the backward of such a call FAILS with R2_ERR_INVALID (the forward's image is then invalid; render the view again with
 * the mode off), and callers that use the returned value as a count must not.  Forwards the chain does not take (first call of a
 * size, debug mode), and all voxelizer calls, wait as before.  mode 0: off (default). */
R2_API void r2_defer_count_control(int mode);
/* out[0] forwards that returned a token, [1] forwards that had to wait because 64 tokens were outstanding, [2] backwards that
 * found the prediction short.  out may be NULL (reset only). */
R2_API void r2_defer_count_stats(long long out[3], int reset);
/* process-wide counts since the last reset: out[0] forwards that took the tile-first chain, [1] forwards that did not (no
 * prediction yet, or beyond its limits), [2] chains enqueued a second time because the prediction fell short, [3] renders
 * repeated with the thin-Gaussian variant, [4] forwards whose prediction was seeded from another Gaussian count (the call after
 * a densification).  out may be NULL (reset only). */
R2_API void r2_tile_first_stats(long long out[5], int reset);

/* Stick-first binning of the voxelizer (csrc/voxel_sticks.hip): grids of more than 64 and up to 32 768 tiles (the 256^3 query of
 * test.py:105-112) are binned without a global sort -- instances are counted and scattered per STICK of up to 8 consecutive
 * tiles (per tile up to 4096 tiles) and every stick's list is sorted on (tile, z bits, id) on its own.  point_list, ranges,
 * volumes and gradients are identical on both chains; a large scene with very long lists (a list of more than 20 480 instances
 * and more than 8 Mi instances in all: the part-wise sort of such lists costs more than the general chain's radix passes;
 * r2_voxel_sticks_limits) continues on the general chain after the preprocess, and the calling thread remembers that for the
 * (P, grid).  Debug mode, larger grids
 * and P >= 2^29 always take the general chain, and so does a device that cannot give a workgroup the chain's 77.8 KB of LDS (parts
 * with a 64 KB limit; gfx950 has 160 KB per CU) -- r2_path_stats reports it as voxel.general.device_lds.  A (P, grid) that handed
 * over is remembered by the calling thread for its next 64 calls, then tried again.  mode 0: never, 1: whenever applicable
 * (default; the environment variable R2_VOXEL_STICKS=0 also switches it off), 3: forget the calling thread's notes; 4 / 5
 * (tests): lists of more than 8192 instances count as unsupported / are sorted in parts (default). */
R2_API void r2_voxel_sticks_control(int mode);
/* The two limits of that rule (process-wide; <= 0: the default).  Every thread's notes are dropped. */
R2_API void r2_voxel_sticks_limits(long long longest_list, long long instances);
/* process-wide counts since the last reset: out[0] forwards that took the chain, [1] forwards that left it after its scan for
 * the general chain, [2] forwards it declined.  out may be NULL (reset only). */
R2_API void r2_voxel_sticks_stats(long long out[3], int reset);

/* Which chain a forward took, and why the others did not (csrc/dispatch.hpp states the rules in one table): process-wide counters
 * since the last reset.  r2_path_stat_count() counters, r2_path_stat_name(i) names them ("raster.tile_first",
 * "raster.general.no_prediction", "voxel.stick_first", "voxel.general.long_lists", "raster.event.second_pass" ...), r2_path_stats
 * copies min(n, count) of them to out (may be NULL) and returns the count.  Results never depend on the chain. */
R2_API int r2_path_stat_count(void);
R2_API const char *r2_path_stat_name(int i);
R2_API int r2_path_stats(long long *out, int n, int reset);

/* Per-thread state.  The library keeps a few KB per host thread: self-resetting device counters of the tile-first rasterizer chain
 * and of the voxelizer's small-grid path and stick-first chain (one block per (device, stream) the thread has used, at most 16 of each:
 * the least recently used one is evicted), 128 bytes of pinned host memory for the num_rendered read-back, and the thread's
 * predictions and notes.
 * These -- and the convenience form r2_knn_dist2 -- are the only memory the library obtains itself; all of it is released when the
 * thread exits, or earlier by this call (waits for the device; the thread's next forward starts over). */
R2_API void r2_thread_release(void);

/* ---- introspection used by the parity tests (bit-exact tile / sort indices) ------------------- */
/* Byte offsets of the private arrays inside the state buffers of a forward call with the given sizes; lets
 * tests read the binning intermediates back without fixing the layout in the ABI.  which:
 *   0 tiles_touched u32[P]      1 point_offsets u32[P] (inclusive scan over Gaussians in depth order; with a depth hint
 *                                 only the entries of the visible Gaussians -- the first nvis -- are written)
 *   2 tiles_unsorted u32[R]     3 values_unsorted u32[R] (emission: depth-ordered Gaussians, tiles y/x-minor)
 *   4 tiles_sorted u32[R] (valid after backward, or for > 4096 tiles)   5 point_list u32[R] (== the reference's sorted point_list)
 *   6 ranges uint2[T]           7 cov3D f32[6P]
 *   8 n_contrib u32[N] (only filled when forward ran with debug != 0)
 *   9 packed render records f32[8P] (voxelizer: f32[12P])                14 {opacity, mu} f32[2P] (rasterizer)
 *  10 depth sort keys u32[P] (bits of the depth; 0xFFFFFFFF for culled Gaussians)
 *  11 first-instance index u32[P]   12 depth order u32[P] (ids sorted by (depth, id); culled ones behind, or unwritten with a hint)
 *  15 host-read words u32[8]: {num_rendered, overflow, thin flag, key extrema x4, nvis}
 *  13 inv u32[R] (sorted position of every emitted instance: the inverse permutation of the tile sort; the voxelizer only
 *     writes it for < 4096 tiles: single-pass tile sort)
 * buffer ids: 0 geometry, 1 binning, 2 image.  Returns -1 for an unknown id. */
R2_API long long r2_raster_state_offset(int which, int P, long long R, int width, int height, int *buffer_id);
R2_API long long r2_voxel_state_offset(int which, int P, long long R, int nx, int ny, int nz, int *buffer_id);

#ifdef __cplusplus
}
#endif
#endif /* R2HIP_H */
