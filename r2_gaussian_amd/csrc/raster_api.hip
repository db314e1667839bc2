// raster_api.hip -- extern "C" entry points of the rasterizer (see include/r2hip.h).
// Host orchestration of Rasterizer::forward / backward (RAS/rasterizer_impl.cu:196-421).
#include "raster_state.hpp"
#include "voxel_state.hpp"
#include "dispatch.hpp"

using namespace r2;

// Forward of V views of the same Gaussians (V = 1: the reference's call).  viewmatrices / projmatrices: [V,16]; out_color
// [V,H,W]; radii [V,P].  See raster_state.hpp (tile_decode) for how the views share one pipeline.
static int raster_forward_impl(
    const char *what, r2_alloc_fn geometryBuffer, void *geometry_user, r2_alloc_fn binningBuffer, void *binning_user,
    r2_alloc_fn imageBuffer, void *image_user, int P, int V, int width, int height, const float *means3D,
    const float *opacities, const float *scales, float scale_modifier, const float *rotations,
    const float *cov3D_precomp, const float *viewmatrices, const float *projmatrices,
    float tan_fovx, float tan_fovy, int mode, float *out_color, int *radii, int debug, hipStream_t s)
{
    host_mark_forward_begin();
    if (P < 0 || V < 1 || width <= 0 || height <= 0 || !geometryBuffer || !binningBuffer || !imageBuffer || !out_color) {
        set_error("%s: invalid argument", what);
        return R2_ERR_INVALID;
    }
    const size_t N = (size_t)width * height * (size_t)V;
    const int gx = (width + TILE2D - 1) / TILE2D, gy = (height + TILE2D - 1) / TILE2D;
    const size_t T = (size_t)gx * gy * (size_t)V;
    if ((size_t)P * (size_t)V >= (size_t)1 << 31 || T >= (size_t)1 << 24) {
        set_error("%s: %d views x %d Gaussians / %zu tiles exceed the 31-bit instance / 24-bit tile range", what, V, P, T);
        return R2_ERR_INVALID;
    }
    if (P == 0) {   // the torch boundary skips the call (SUB/rasterize_points.cu:70); out_color is pre-zeroed
        R2_HIP_TRY(hipMemsetAsync(out_color, 0, N * sizeof(float), s));
        return 0;
    }
    if (!means3D || !opacities || !viewmatrices || !projmatrices || !radii ||
        (!cov3D_precomp && (!scales || !rotations))) {
        set_error("%s: NULL input (need means3D, opacities, matrices, radii and scales+rotations or cov3D_precomp)", what);
        return R2_ERR_INVALID;
    }
    if (mode != 0 && mode != 1) {
        set_error("%s: unsupported mode %d", what, mode);
        return R2_ERR_INVALID;
    }
    const int PV = P * V;   // view instances: everything between the preprocess and the render kernels works on these

    // tile-first binning (raster_tilefirst.hip): calls whose instance count this thread can predict from its recent calls (one
    // view, or V stacked views of up to 4096 tiles in all); everything else -- first call of a size, debug mode, huge grids --
    // takes the chain below
    if (!debug) {
        const int r = raster_forward_tilefirst(what, geometryBuffer, geometry_user, binningBuffer, binning_user, imageBuffer, image_user,
                                               P, V, width, height, means3D, opacities, scales, scale_modifier, rotations, cov3D_precomp,
                                               viewmatrices, projmatrices, tan_fovx, tan_fovy, mode, out_color, radii, s);
        if (r != TF_NOT_TAKEN) {
            if (r >= 0) host_mark_forward_end();
            return r;
        }
    } else {
        path_count(PS_RAS_GENERAL_DEBUG);
    }

    char *gchunk = geometryBuffer(RasterGeom::carve(nullptr, PV).bytes, geometry_user);
    if (!gchunk) {
        set_error("%s: state allocation callback returned NULL", what);
        return R2_ERR_ALLOC;
    }
    const RasterGeom geom = RasterGeom::carve(gchunk, PV);

    // Binning, first half: view instances in (depth, id) order + their instance offsets in that order.
    //   hinted path (depth range known from the previous call with this size): preprocess registers every key in its bucket,
    //   one dual scan + place + rank finish the job, and the host's read-back overlaps place + rank;
    //   un-hinted path: min/max, count, scan, place, rank, then the offsets scan;
    //   either may overflow a bucket (many identical depths / a stale hint): general radix sort + scan instead.
    int rc = depth_order_prepare(geom.dorder_temp, geom.dorder_bytes, (size_t)PV, s);   // zeroes counters + the host-read words
    if (rc) return rc;
    uint32_t *host_words = geom.host_words;
    DepthHint hint;
    const bool hinted = depth_hint_lookup(0, (size_t)PV, &hint);
    const uint32_t pre_wgs = (uint32_t)((PV + 255) / 256);
    DepthReg reg{};
    // sorted records for the emission kernel: the preprocess hands each key's tile rectangle (one word) to the depth order
    static const bool rects_on = [] { const char *e = getenv("R2_SORTED_RECORDS"); return !(e && e[0] == '0'); }();
    const bool rects = hinted && rects_on && gx <= 256 && gy <= 256;
    if (hinted) reg = depth_order_reg(geom.dorder_temp, (size_t)PV, hint, rects);
    { StageScope t(ST_RAS_PREPROCESS, s);
    launch_raster_preprocess(geom, P, V, means3D, scales, scale_modifier, rotations, opacities, cov3D_precomp, viewmatrices,
                             projmatrices, width, height, tan_fovx, tan_fovy, mode, radii, host_words + DW_USER, reg, true, s); }
    R2_STAGE_CHECK(debug, s, "preprocess");
    uint32_t hw[DW_COUNT] = { 0 };
    if (hinted) {
        uint32_t *mailbox = nullptr, mailbox_seq = 0;
        rc = host_mailbox_arm(&mailbox, &mailbox_seq);
        if (rc) return rc;
        { StageScope t(ST_RAS_SCAN, s);
        rc = depth_order_fast_scan(geom.dorder_temp, (size_t)PV, pre_wgs, mailbox, mailbox_seq, s); }
        if (rc) return rc;
        { StageScope t(ST_RAS_DEPTHSORT, s);
        rc = depth_order_fast_finish(geom.dorder_temp, (size_t)PV, geom.depth_key, geom.tiles_touched, geom.order, geom.offsets, s, rects); }
        if (rc) return rc;
        rc = host_mailbox_wait(mailbox_seq, hw, DW_COUNT, s);   // the GPU places + ranks while the host waits for the words
        if (rc) return rc;
        R2_STAGE_CHECK(debug, s, "depth order (hinted)");
    } else {
        { StageScope t(ST_RAS_DEPTHSORT, s);
        rc = depth_order_buckets(geom.dorder_temp, geom.dorder_bytes, geom.depth_key, geom.order, (size_t)PV, s); }
        if (rc) return rc;
        R2_STAGE_CHECK(debug, s, "depth order");
        // (fusing the scan's per-group reduction into the depth order's last kernel with per-wave atomics was measured
        // 4x slower than this separate 5 us kernel: ~5k atomics on ~75 addresses serialise at the memory side)
        { StageScope t(ST_RAS_SCAN, s);
        rc = inclusive_scan_gather_u32(geom.scan_temp, geom.scan_bytes, geom.tiles_touched, geom.order, geom.offsets, PV, s,
                                       host_words + DW_TOTAL); }
        if (rc) return rc;
        R2_STAGE_CHECK(debug, s, "scan");
        // total number of (tile, Gaussian) instances: sizes the binning state (the reference's D2H, RAS/rasterizer_impl.cu:279)
        rc = read_host_words(host_words, hw, DW_COUNT, s);
        if (rc) return rc;
    }
    uint32_t num_rendered = hw[DW_TOTAL];
    const bool overflow = hw[DW_OVERFLOW] != 0;
    bool full_order = !hinted;   // order / offsets cover all view instances (else only the visible prefix)
    if (overflow) {   // general radix sort instead
        rc = sort_pairs_ex(geom.psort_temp, geom.psort_bytes, geom.depth_key, geom.depth_sorted, nullptr /* values = indices */, geom.order, nullptr,
                           nullptr, (size_t)PV, 32, /*allow_skip=*/true, nullptr, s);
        if (!rc) rc = inclusive_scan_gather_u32(geom.scan_temp, geom.scan_bytes, geom.tiles_touched, geom.order, geom.offsets, PV,
                                                s, host_words + DW_TOTAL);
        uint32_t total = 0;
        if (!rc) rc = read_host_words(host_words + DW_TOTAL, &total, 1, s);
        if (rc) return rc;
        num_rendered = total;
        full_order = true;
    }
    depth_hint_update(0, (size_t)PV, hw, overflow);
    if (num_rendered > 0x7FFFFFFFu) {   // the API returns it as a non-negative int (like the reference's int num_rendered)
        set_error("%s: more than 2147483647 (tile, Gaussian) instances: they do not fit the 31-bit num_rendered", what);
        return R2_ERR_INVALID;
    }
    const size_t R = num_rendered;

    // both remaining state buffers are sized by R: the sorted lists (+ backward scratch) and the
    // forward work list / per-chunk partial images
    char *bchunk = binningBuffer(RasterBinning::carve(nullptr, R).bytes, binning_user);
    char *ichunk = imageBuffer(RasterImage::carve(nullptr, T, N, R, debug != 0).bytes, image_user);
    if (!bchunk || !ichunk) {
        set_error("%s: binning/image allocation callback returned NULL", what);
        return R2_ERR_ALLOC;
    }
    const RasterBinning bin = RasterBinning::carve(bchunk, R);
    const RasterImage img = RasterImage::carve(ichunk, T, N, R, debug != 0);

    const uint32_t *tile_counts = nullptr;
    bool work_built = false;   // ranges + work list already produced by the sort's last kernel
    if (R > 0) {
        const int bit = (int)higher_msb((uint32_t)(T > 1 ? T - 1 : 1));   // bits of the largest tile id (the reference
                                                                     // sorts getHigherMsb(T) bits: one more for T = 2^k)
        bool hist_ready = false;   // the emission kernel built the tile sort's histograms as well
        { StageScope t(ST_RAS_DUPLICATE, s);
        if (rects && !full_order) {
            static const bool fuse_on = [] { const char *e = getenv("R2_EMIT_HIST"); return !(e && e[0] == '0'); }();
            TileSortPlan plan;
            if (fuse_on && !debug && tile_sort_plan(bin.sort_temp, bin.sort_bytes, R, bit, &plan))
                hist_ready = launch_raster_emit_hist(geom, bin, P, V, width, height, host_words + DW_NVIS, R, plan, s);
            if (!hist_ready) launch_raster_duplicate_sorted(geom, bin, P, V, width, height, host_words + DW_NVIS, s);
        } else {
            launch_raster_duplicate(geom, bin, P, V, radii, width, height, full_order ? nullptr : host_words + DW_NVIS, s);
        } }
        R2_STAGE_CHECK(debug, s, "duplicateWithKeys");
        // stable sort by tile id; payloads: the emission index (-> perm, the backward's scratch row) and the Gaussian
        // id (-> point_list)
        { StageScope t(ST_RAS_SORT, s);
        if (sort_is_single_pass(bit)) {
            const WorkListOut wo{img.ranges, img.chunk_base, img.work_tile, (uint32_t)T, FWD_CHUNK,
                                 debug ? nullptr : img.tile_done, 0u, 0u,
                                 (raster_forward_wave_kernel_on() && raster_ids_leave_room_for_masks((size_t)PV)) ? 1u : 0u};
            rc = sort_by_tile_single_pass(bin.sort_temp, bin.sort_bytes, bin.tiles_unsorted, bin.vals_unsorted, bin.point_list,
                                          debug ? bin.inv : nullptr, R, bit, &tile_counts, s, &wo, hist_ready);   // inv: introspection only
            work_built = true;
        } else {   // > 4096 tiles: general multi-pass sort, then invert its permutation (the scratch is free until backward)
            uint32_t *perm = reinterpret_cast<uint32_t *>(bin.part);   // scratch for the intermediate pass (free until backward)
            rc = sort_pairs_ex(bin.sort_temp, bin.sort_bytes, bin.tiles_unsorted, bin.tiles, nullptr, perm, bin.vals_unsorted,
                               bin.point_list, R, bit, false, nullptr, s, bin.inv);
        } }
        if (rc) return rc;
        R2_STAGE_CHECK(debug, s, "sort");
    }
    { StageScope t(ST_RAS_RANGES, s);
    if (work_built) {
        // nothing to do
    } else if (tile_counts) {   // single-pass sort: per-tile counts are a by-product
        launch_ranges_and_work(tile_counts, (uint32_t)T, FWD_CHUNK, img.ranges, img.chunk_base, img.work_tile, s);
    } else {
        rc = tile_ranges(bin.tiles, nullptr, nullptr, nullptr, R, img.ranges, T, s);
        if (rc) return rc;
        launch_build_work(img.ranges, (uint32_t)T, FWD_CHUNK, img.chunk_base, img.work_tile, img.work_temp, s);
    } }
    R2_STAGE_CHECK(debug, s, "identifyTileRanges");
    { StageScope t(ST_RAS_RENDER_FWD, s);
    // single-pass sort: the tile's last work item (or the combine kernel) also writes tiles[k] for the backward
    launch_raster_render_forward(geom, bin, img, width, height, V, out_color, debug != 0, tile_counts ? bin.tiles : nullptr,
                                 /*fused_combine=*/work_built && debug == 0, s, nullptr, nullptr, (size_t)PV); }
    R2_STAGE_CHECK(debug, s, "render");
    // the next call's prediction (visible keys are positive floats: their range is the P class of the host words)
    raster_tilefirst_note(P, V, width, height, num_rendered, hw[DW_USER] != 0, hw[DW_PMAX], ~hw[DW_PNMAX]);
    host_mark_forward_end();
    return (int)num_rendered;
}

static int raster_backward_impl(
    const char *what, int P, int V, int R, int width, int height, const float *means3D, const float *scales, float scale_modifier,
    const float *rotations, const float *cov3D_precomp, const float *viewmatrices, const float *projmatrices,
    float tan_fovx, float tan_fovy, const int *radii, char *geom_buffer, char *binning_buffer,
    const float *dL_dpix, float *dL_dmean2D, float *dL_dconic, float *dL_dopacity, float *dL_dmu,
    float *dL_dmean3D, float *dL_dcov3D, float *dL_dscale, float *dL_drot, int mode, int debug, hipStream_t s)
{
    if (P == 0) return 0;
    if (P < 0 || V < 1 || R < 0 || !means3D || !radii || !geom_buffer || (R > 0 && !binning_buffer) || !dL_dpix ||
        !viewmatrices || !projmatrices || !dL_dmean2D || !dL_dconic || !dL_dopacity || !dL_dmu || !dL_dmean3D || !dL_dcov3D ||
        (!cov3D_precomp && (!scales || !rotations || !dL_dscale || !dL_drot))) {
        set_error("%s: invalid argument", what);
        return R2_ERR_INVALID;
    }
    if (R >= DEFER_TOKEN_FLAG) {   // a deferred forward's token (r2_defer_count_control): the count is on its way, or long here
        uint32_t true_R = 0;
        const int rc = raster_resolve_deferred(what, R, s, &true_R);
        if (rc) return rc;
        R = (int)true_R;
    }
    const RasterGeom geom = RasterGeom::carve(geom_buffer, P * V);
    const RasterBinning bin = RasterBinning::carve(binning_buffer, (size_t)R);
    { StageScope t(ST_RAS_RENDER_BWD, s);
    launch_raster_render_backward(geom, bin, radii, width, height, V, (size_t)R, dL_dpix, s, (size_t)P * (size_t)V); }
    R2_STAGE_CHECK(debug, s, "render backward");
    const float *cov3D = cov3D_precomp ? cov3D_precomp : geom.cov3D;
    { StageScope t(ST_RAS_GEOM_BWD, s);
    launch_raster_geom_backward(P, V, means3D, radii, cov3D, scales, rotations, scale_modifier, width, height, tan_fovx,
                                tan_fovy, viewmatrices, projmatrices, dL_dconic, dL_dmu, dL_dmean2D, dL_dopacity, dL_dmean3D,
                                dL_dcov3D, dL_dscale, dL_drot, mode, geom, bin.part, s); }
    R2_STAGE_CHECK(debug, s, "geometry backward");
    return 0;
}

extern "C" int r2_raster_forward(
    r2_alloc_fn geometryBuffer, void *geometry_user, r2_alloc_fn binningBuffer, void *binning_user,
    r2_alloc_fn imageBuffer, void *image_user, int P, int width, int height, const float *means3D,
    const float *opacities, const float *scales, float scale_modifier, const float *rotations,
    const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix, const float *cam_pos,
    float tan_fovx, float tan_fovy, int prefiltered, int mode, float *out_color, int *radii, int debug, void *stream)
{
    (void)cam_pos;
    (void)prefiltered;   // the reference only uses it to trap on an impossible state (RAS/auxiliary.h:160-164)
    return raster_forward_impl("r2_raster_forward", geometryBuffer, geometry_user, binningBuffer, binning_user, imageBuffer, image_user,
                               P, 1, width, height, means3D, opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix,
                               projmatrix, tan_fovx, tan_fovy, mode, out_color, radii, debug, (hipStream_t)stream);
}

extern "C" int r2_raster_backward(
    int P, int R, int width, int height, const float *means3D, const float *scales, float scale_modifier,
    const float *rotations, const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix,
    const float *campos, float tan_fovx, float tan_fovy, const int *radii, char *geom_buffer, char *binning_buffer,
    char *img_buffer, const float *dL_dpix, float *dL_dmean2D, float *dL_dconic, float *dL_dopacity, float *dL_dmu,
    float *dL_dmean3D, float *dL_dcov3D, float *dL_dscale, float *dL_drot, int mode, int debug, void *stream)
{
    (void)campos;
    (void)img_buffer;
    return raster_backward_impl("r2_raster_backward", P, 1, R, width, height, means3D, scales, scale_modifier, rotations, cov3D_precomp,
                                viewmatrix, projmatrix, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, dL_dpix, dL_dmean2D,
                                dL_dconic, dL_dopacity, dL_dmu, dL_dmean3D, dL_dcov3D, dL_dscale, dL_drot, mode, debug,
                                (hipStream_t)stream);
}

// ---- batched views (new functionality; the reference renders one view per call): see include/r2hip.h
extern "C" int r2_raster_forward_batch(
    r2_alloc_fn geometryBuffer, void *geometry_user, r2_alloc_fn binningBuffer, void *binning_user,
    r2_alloc_fn imageBuffer, void *image_user, int P, int V, int width, int height, const float *means3D,
    const float *opacities, const float *scales, float scale_modifier, const float *rotations,
    const float *cov3D_precomp, const float *viewmatrices, const float *projmatrices,
    float tan_fovx, float tan_fovy, int mode, float *out_color, int *radii, int debug, void *stream)
{
    return raster_forward_impl("r2_raster_forward_batch", geometryBuffer, geometry_user, binningBuffer, binning_user, imageBuffer,
                               image_user, P, V, width, height, means3D, opacities, scales, scale_modifier, rotations, cov3D_precomp,
                               viewmatrices, projmatrices, tan_fovx, tan_fovy, mode, out_color, radii, debug, (hipStream_t)stream);
}

extern "C" int r2_raster_backward_batch(
    int P, int V, int R, int width, int height, const float *means3D, const float *scales, float scale_modifier,
    const float *rotations, const float *cov3D_precomp, const float *viewmatrices, const float *projmatrices,
    float tan_fovx, float tan_fovy, const int *radii, char *geom_buffer, char *binning_buffer, char *img_buffer,
    const float *dL_dpix, float *dL_dmean2D, float *dL_dconic, float *dL_dopacity, float *dL_dmu,
    float *dL_dmean3D, float *dL_dcov3D, float *dL_dscale, float *dL_drot, int mode, int debug, void *stream)
{
    (void)img_buffer;
    return raster_backward_impl("r2_raster_backward_batch", P, V, R, width, height, means3D, scales, scale_modifier, rotations,
                                cov3D_precomp, viewmatrices, projmatrices, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer,
                                dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, dL_dmu, dL_dmean3D, dL_dcov3D, dL_dscale, dL_drot, mode,
                                debug, (hipStream_t)stream);
}

extern "C" int r2_mark_visible(int P, const float *means3D, const float *viewmatrix, const float *projmatrix,
                               uint8_t *present, void *stream)
{
    (void)projmatrix;
    if (P == 0) return 0;
    if (P < 0 || !means3D || !viewmatrix || !present) {
        set_error("r2_mark_visible: invalid argument");
        return R2_ERR_INVALID;
    }
    launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
    R2_STAGE_CHECK(0, (hipStream_t)stream, "markVisible");
    return 0;
}

extern "C" long long r2_raster_state_offset(int which, int P, long long R, int width, int height, int *buffer_id)
{
    char *const base = reinterpret_cast<char *>(uintptr_t(1) << 40);   // fake base: only differences are used
    const int gx = (width + TILE2D - 1) / TILE2D, gy = (height + TILE2D - 1) / TILE2D;
    const RasterGeom g = RasterGeom::carve(base, P);
    const RasterBinning b = RasterBinning::carve(base, (size_t)R);
    const RasterImage im = RasterImage::carve(base, (size_t)gx * gy, (size_t)width * height, (size_t)R, true);
    const char *p = nullptr;
    int buf = -1;
    switch (which) {
    case 0: p = (char *)g.tiles_touched; buf = 0; break;
    case 1: p = (char *)g.offsets; buf = 0; break;
    case 2: p = (char *)b.tiles_unsorted; buf = 1; break;
    case 3: p = (char *)b.vals_unsorted; buf = 1; break;
    case 4: p = (char *)b.tiles; buf = 1; break;
    case 5: p = (char *)b.point_list; buf = 1; break;
    case 6: p = (char *)im.ranges; buf = 2; break;
    case 7: p = (char *)g.cov3D; buf = 0; break;
    case 8: p = (char *)im.n_contrib; buf = 2; break;
    case 9: p = (char *)g.rec; buf = 0; break;
    case 10: p = (char *)g.depth_key; buf = 0; break;
    case 12: p = (char *)g.order; buf = 0; break;
    case 11: p = (char *)g.first; buf = 0; break;
    case 13: p = (char *)b.inv; buf = 1; break;
    case 14: p = (char *)g.op_mu; buf = 0; break;
    case 15: p = (char *)g.host_words; buf = 0; break;
    default: return -1;
    }
    if (buffer_id) *buffer_id = buf;
    return (long long)(p - base);
}

// Everything the library keeps per host thread -- the self-resetting counter blocks of the tile-first rasterizer chain and of the
// small-grid voxelizer path (a few KB of device memory per (device, stream) the thread has used), its pinned mailbox words, its
// predictions -- is released when the thread exits; a long-lived thread can give it back earlier with this call.  Waits for
// the device (hipFree).  The next forward of the thread simply starts over.
extern "C" void r2_thread_release(void)
{
    r2::raster_tilefirst_release();
    r2::voxel_small_release();
    r2::voxel_sticks_release();
    r2::host_words_release();
}
