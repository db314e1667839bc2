// voxel_api.hip -- extern "C" entry points of the voxelizer (see include/r2hip.h).
// Host orchestration of Voxelizer::forward / backward (VOX/voxelizer_impl.cu:171-389).
#include "voxel_state.hpp"
#include "dispatch.hpp"

using namespace r2;

// tile layers [tile_x0, tile_x1) along x of the full grid (an ordinary call: all of them, tile_x1 < 0)
static VoxelGrid make_grid(int nx, int ny, int nz, float sx, float sy, float sz, float cx, float cy, float cz, int tile_x0 = 0,
                           int tile_x1 = -1)
{
    VoxelGrid v;
    v.ny = ny; v.nz = nz;
    v.sx = sx; v.sy = sy; v.sz = sz;
    v.cx = cx; v.cy = cy; v.cz = cz;
    v.fnx = nx;
    v.fgx = (nx + TILE3D - 1) / TILE3D;
    if (tile_x1 < 0) tile_x1 = v.fgx;
    v.gx = tile_x1 - tile_x0;
    v.ox = tile_x0 * TILE3D;
    v.nx = (tile_x1 * TILE3D < nx ? tile_x1 * TILE3D : nx) - v.ox;   // the last layer of the full grid may be ragged
    v.gy = (ny + TILE3D - 1) / TILE3D;
    v.gz = (nz + TILE3D - 1) / TILE3D;
    return v;
}
static bool slab_ok(int nx, int tile_x0, int tile_x1)
{
    return tile_x0 >= 0 && tile_x1 > tile_x0 && tile_x1 <= (nx + TILE3D - 1) / TILE3D;
}

extern "C" int r2_voxel_forward(
    r2_alloc_fn geometryBuffer, void *geometry_user, r2_alloc_fn binningBuffer, void *binning_user,
    r2_alloc_fn imageBuffer, void *image_user, int P, int nVoxel_x, int nVoxel_y, int nVoxel_z, float sVoxel_x,
    float sVoxel_y, float sVoxel_z, float center_x, float center_y, float center_z, const float *means3D,
    const float *opacities, const float *scales, float scale_modifier, const float *rotations,
    const float *cov3D_precomp, int prefiltered, float *out_volume, int *radii_x, int *radii_y, int *radii_z,
    int debug, void *stream)
{
    return r2_voxel_forward_slab(geometryBuffer, geometry_user, binningBuffer, binning_user, imageBuffer, image_user, P, nVoxel_x,
                                 nVoxel_y, nVoxel_z, sVoxel_x, sVoxel_y, sVoxel_z, center_x, center_y, center_z, 0,
                                 nVoxel_x > 0 ? (nVoxel_x + TILE3D - 1) / TILE3D : 1, means3D, opacities, scales, scale_modifier,
                                 rotations, cov3D_precomp, prefiltered, out_volume, radii_x, radii_y, radii_z, debug, stream);
}

extern "C" int r2_voxel_forward_slab(
    r2_alloc_fn geometryBuffer, void *geometry_user, r2_alloc_fn binningBuffer, void *binning_user,
    r2_alloc_fn imageBuffer, void *image_user, int P, int nVoxel_x, int nVoxel_y, int nVoxel_z, float sVoxel_x,
    float sVoxel_y, float sVoxel_z, float center_x, float center_y, float center_z, int tile_x0, int tile_x1,
    const float *means3D, const float *opacities, const float *scales, float scale_modifier, const float *rotations,
    const float *cov3D_precomp, int prefiltered, float *out_volume, int *radii_x, int *radii_y, int *radii_z,
    int debug, void *stream)
{
    (void)prefiltered;
    hipStream_t s = (hipStream_t)stream;
    host_mark_forward_begin();
    if (P < 0 || nVoxel_x <= 0 || nVoxel_y <= 0 || nVoxel_z <= 0 || !geometryBuffer || !binningBuffer ||
        !imageBuffer || !out_volume || !slab_ok(nVoxel_x, tile_x0, tile_x1)) {
        set_error("r2_voxel_forward: invalid argument");
        return R2_ERR_INVALID;
    }
    const VoxelGrid v = make_grid(nVoxel_x, nVoxel_y, nVoxel_z, sVoxel_x, sVoxel_y, sVoxel_z, center_x, center_y, center_z, tile_x0, tile_x1);
    const size_t V = (size_t)v.nx * nVoxel_y * nVoxel_z;   // the output block: the slab's voxels
    const size_t T = (size_t)v.gx * v.gy * v.gz;
    if (P == 0) {
        R2_HIP_TRY(hipMemsetAsync(out_volume, 0, V * sizeof(float), s));
        return 0;
    }
    // The voxel radius is derived from the raw scales even when a precomputed covariance is given
    // (VOX/forward.cu:137-143): the reference dereferences an empty tensor there; we refuse instead.
    if (!means3D || !opacities || !radii_x || !radii_y || !radii_z || !scales || (!cov3D_precomp && !rotations)) {
        set_error("r2_voxel_forward: NULL input (scales are required even with cov3D_precomp)");
        return R2_ERR_INVALID;
    }
    char *gchunk = geometryBuffer(VoxelGeom::carve(nullptr, P).bytes, geometry_user);
    if (!gchunk) {
        set_error("r2_voxel_forward: state allocation callback returned NULL");
        return R2_ERR_ALLOC;
    }
    const VoxelGeom geom = VoxelGeom::carve(gchunk, P);

    // which chain (dispatch.hpp): small grids (the 32^3 TV patch of the training loop: survivors only, 4 launches, voxel_small.hip),
    // grids of 65 to 32 768 tiles (64^3 ... the 256^3 query: stick-first binning, no global sort, voxel_sticks.hip), or the
    // general chain below -- also where the other two hand over to when the CALL turns out to be beyond them
    bool preprocessed = false;   // the stick chain left after its scan (a list too long for it): the preprocess has run, un-hinted
    const VoxelChoice choice = voxel_forward_choice(v, (size_t)P, debug != 0, voxel_small_switched_on(), voxel_small_lds_ok(),
                                                    voxel_sticks_switched_on(), voxel_sticks_lds_ok());
    if (choice.chain == VOX_CHAIN_SMALL) {
        const int small = voxel_forward_small(binningBuffer, binning_user, imageBuffer, image_user, geom, v, P, means3D, opacities,
                                              scales, scale_modifier, rotations, cov3D_precomp, out_volume, radii_x, radii_y, radii_z, s);
        if (small != VOX_SMALL_NOT_TAKEN) {
            host_mark_forward_end();
            return small;
        }
    } else if (choice.chain == VOX_CHAIN_STICKS) {
        const int st = voxel_forward_sticks(binningBuffer, binning_user, imageBuffer, image_user, geom, v, P, means3D, opacities, scales,
                                            scale_modifier, rotations, cov3D_precomp, out_volume, radii_x, radii_y, radii_z,
                                            choice.stick_shift, s);
        if (st == VOX_STICKS_FALLBACK) {
            preprocessed = true;
        } else if (st != VOX_STICKS_NOT_TAKEN) {
            host_mark_forward_end();
            return st;
        }
    } else {
        path_count(choice.why);
    }

    // binning, first half (see raster_api.hip): order of the Gaussians by the bits of world z (the reference's low sort
    // word, Q10) + instance offsets in that order; hinted / un-hinted bucket sort, radix fallback
    int rc = depth_order_prepare(geom.dorder_temp, geom.dorder_bytes, (size_t)P, s);
    if (rc) return rc;
    uint32_t *host_words = geom.host_words;
    DepthHint hint;
    const bool hinted = !preprocessed && depth_hint_lookup(1, (size_t)P, &hint);
    const uint32_t pre_wgs = (uint32_t)((P + 255) / 256);
    DepthReg reg{};
    if (hinted) reg = depth_order_reg(geom.dorder_temp, (size_t)P, hint);
    if (!preprocessed) {
        StageScope t(ST_VOX_PREPROCESS, s);
        launch_voxel_preprocess(geom, v, P, means3D, scales, scale_modifier, rotations, opacities, cov3D_precomp, radii_x,
                                radii_y, radii_z, reg, true, s);
    }
    R2_STAGE_CHECK(debug, s, "preprocess");
    uint32_t hw[DW_COUNT] = { 0 };
    if (hinted) {
        uint32_t *mailbox = nullptr, mailbox_seq = 0;
        rc = host_mailbox_arm(&mailbox, &mailbox_seq);
        if (rc) return rc;
        { StageScope t(ST_VOX_SCAN, s);
        rc = depth_order_fast_scan(geom.dorder_temp, (size_t)P, pre_wgs, mailbox, mailbox_seq, s); }
        if (rc) return rc;
        { StageScope t(ST_VOX_DEPTHSORT, s);
        rc = depth_order_fast_finish(geom.dorder_temp, (size_t)P, geom.depth_key, geom.tiles_touched, geom.order, geom.offsets, s); }
        if (rc) return rc;
        rc = host_mailbox_wait(mailbox_seq, hw, DW_COUNT, s);   // the GPU places + ranks while the host waits for the words
        if (rc) return rc;
        R2_STAGE_CHECK(debug, s, "depth order (hinted)");
    } else {
        { StageScope t(ST_VOX_DEPTHSORT, s);
        rc = depth_order_buckets(geom.dorder_temp, geom.dorder_bytes, geom.depth_key, geom.order, (size_t)P, s); }
        if (rc) return rc;
        R2_STAGE_CHECK(debug, s, "depth order");
        { StageScope t(ST_VOX_SCAN, s);
        rc = inclusive_scan_gather_u32(geom.scan_temp, geom.scan_bytes, geom.tiles_touched, geom.order, geom.offsets, P, s,
                                       host_words + DW_TOTAL); }
        if (rc) return rc;
        R2_STAGE_CHECK(debug, s, "scan");
        // total number of (tile, Gaussian) instances: sizes the binning state (the reference's D2H, VOX/voxelizer_impl.cu:248)
        rc = read_host_words(host_words, hw, DW_COUNT, s);
        if (rc) return rc;
    }
    uint32_t num_rendered = hw[DW_TOTAL];
    const bool overflow = hw[DW_OVERFLOW] != 0;
    bool full_order = !hinted;   // order / offsets cover all P Gaussians (else only the visible prefix)
    if (overflow) {   // general radix sort instead
        rc = sort_pairs_ex(geom.psort_temp, geom.psort_bytes, geom.depth_key, geom.depth_sorted, nullptr /* values = indices */, geom.order, nullptr,
                           nullptr, (size_t)P, 32, /*allow_skip=*/true, nullptr, s);
        if (!rc) rc = inclusive_scan_gather_u32(geom.scan_temp, geom.scan_bytes, geom.tiles_touched, geom.order, geom.offsets, P,
                                                s, host_words + DW_TOTAL);
        uint32_t total = 0;
        if (!rc) rc = read_host_words(host_words + DW_TOTAL, &total, 1, s);
        if (rc) return rc;
        num_rendered = total;
        full_order = true;
        // `order` is now a permutation of ALL P ids (culled ones interleaved: their depth_key is bits(z), not a sentinel),
        // but the dual scan left its visible count in the device word: the geometry backward would walk only that prefix
        // and skip visible Gaussians sorted behind it.  0 = "walk all of it" (voxel_geom_backward_kernel).
        R2_HIP_TRY(hipMemsetAsync(host_words + DW_NVIS, 0, sizeof(uint32_t), s));
    }
    depth_hint_update(1, (size_t)P, hw, overflow);
    if (num_rendered > 0x7FFFFFFFu) {   // the API returns it as a non-negative int (like the reference's int num_rendered)
        set_error("r2_voxel_forward: %u (tile, Gaussian) instances do not fit the 31-bit num_rendered", num_rendered);
        return R2_ERR_INVALID;
    }
    const size_t R = num_rendered;

    char *bchunk = binningBuffer(VoxelBinning::carve(nullptr, R).bytes, binning_user);
    char *ichunk = imageBuffer(VoxelImage::carve(nullptr, T, V, R, debug != 0, vox_chunk_for(v.gy, v.gz)).bytes, image_user);
    if (!bchunk || !ichunk) {
        set_error("r2_voxel_forward: binning/image allocation callback returned NULL");
        return R2_ERR_ALLOC;
    }
    const VoxelBinning bin = VoxelBinning::carve(bchunk, R);
    const VoxelImage img = VoxelImage::carve(ichunk, T, V, R, debug != 0, vox_chunk_for(v.gy, v.gz));
    const uint32_t *tile_counts = nullptr;
    bool work_built = false;   // ranges + work list already produced by the sort's last kernel
    bool ranges_zeroed = false;   // img.ranges zero-filled by the duplicate kernel
    if (R > 0) {
        { StageScope t(ST_VOX_DUPLICATE, s);
        const int bit0 = (int)higher_msb((uint32_t)(T > 1 ? T - 1 : 1));
        ranges_zeroed = !sort_is_single_pass(bit0);   // the multi-pass path below builds the ranges with tile_ranges()
        launch_voxel_duplicate(geom, bin, v, P, radii_x, radii_y, radii_z, full_order ? nullptr : host_words + DW_NVIS, s,
                               ranges_zeroed ? img.ranges : nullptr, T); }
        R2_STAGE_CHECK(debug, s, "duplicateWithKeys");
        const int bit = (int)higher_msb((uint32_t)(T > 1 ? T - 1 : 1));   // bits of the largest tile id (the reference
                                                                     // sorts getHigherMsb(T) bits: one more for T = 2^k)
        { StageScope t(ST_VOX_SORT, s);
        if (sort_is_single_pass(bit)) {   // block 0 of the sort's last kernel also builds tile ranges + work list
            const WorkListOut wo{img.ranges, img.chunk_base, img.work_tile, (uint32_t)T, vox_work_chunk(v.gy, v.gz), nullptr,
                                 voxel_short_list_min(debug != 0)};
            rc = sort_by_tile_single_pass(bin.sort_temp, bin.sort_bytes, bin.tiles_unsorted, bin.vals_unsorted, bin.point_list,
                                          debug ? bin.inv : nullptr, R, bit, &tile_counts, s, &wo);   // inv: introspection only
            work_built = true;
        } else {   // > 4096 tiles (e.g. 256^3): general multi-pass sort of (tile, Gaussian id) pairs -- no permutation is
                   // carried along: the backward recomputes an instance's emission index from the Gaussian's tile cube
            rc = sort_pairs_ex(bin.sort_temp, bin.sort_bytes, bin.tiles_unsorted, bin.tiles, bin.vals_unsorted, bin.point_list,
                               nullptr, nullptr, R, bit, false, nullptr, s);
        } }
        if (rc) return rc;
        R2_STAGE_CHECK(debug, s, "sort");
    }
    { StageScope t(ST_VOX_RANGES, s);
    if (work_built) {
        // nothing to do
    } else if (tile_counts) {
        launch_ranges_and_work(tile_counts, (uint32_t)T, vox_work_chunk(v.gy, v.gz), img.ranges, img.chunk_base, img.work_tile, s,
                               voxel_short_list_min(debug != 0));
    } else {
        rc = tile_ranges(bin.tiles, nullptr, nullptr, nullptr, R, img.ranges, T, s, ranges_zeroed);
        if (rc) return rc;
        launch_build_work(img.ranges, (uint32_t)T, vox_work_chunk(v.gy, v.gz), img.chunk_base, img.work_tile, img.work_temp, s,
                          voxel_short_list_min(debug != 0));
    } }
    R2_STAGE_CHECK(debug, s, "identifyTileRanges");
    { StageScope t(ST_VOX_RENDER_FWD, s);
    launch_voxel_render_forward(geom, bin, img, v, out_volume, debug != 0, s); }
    R2_STAGE_CHECK(debug, s, "render");
    if (R > 0 && voxel_forward_fills_tiles(T)) fill_tiles_from_ranges(img.ranges, T, bin.tiles, s);   // see voxel_state.hpp
    host_mark_forward_end();
    return (int)num_rendered;
}

extern "C" int r2_voxel_backward(
    int P, int R, int nVoxel_x, int nVoxel_y, int nVoxel_z, float sVoxel_x, float sVoxel_y, float sVoxel_z,
    float center_x, float center_y, float center_z, const float *means3D, const float *scales, float scale_modifier,
    const float *rotations, const float *cov3D_precomp, const int *radii_x, const int *radii_y, const int *radii_z,
    char *geom_buffer, char *binning_buffer, char *img_buffer, const float *dL_dvol, float *dL_dmean3D_norm,
    float *dL_dconic3D, float *dL_dopacity, float *dL_dmean3D, float *dL_dcov3D, float *dL_dscale, float *dL_drot,
    int debug, void *stream)
{
    return r2_voxel_backward_slab(P, R, nVoxel_x, nVoxel_y, nVoxel_z, sVoxel_x, sVoxel_y, sVoxel_z, center_x, center_y, center_z, 0,
                                  nVoxel_x > 0 ? (nVoxel_x + TILE3D - 1) / TILE3D : 1, means3D, scales, scale_modifier, rotations,
                                  cov3D_precomp, radii_x, radii_y, radii_z, geom_buffer, binning_buffer, img_buffer, dL_dvol,
                                  dL_dmean3D_norm, dL_dconic3D, dL_dopacity, dL_dmean3D, dL_dcov3D, dL_dscale, dL_drot, debug, stream);
}

extern "C" int r2_voxel_backward_slab(
    int P, int R, int nVoxel_x, int nVoxel_y, int nVoxel_z, float sVoxel_x, float sVoxel_y, float sVoxel_z,
    float center_x, float center_y, float center_z, int tile_x0, int tile_x1, const float *means3D, const float *scales,
    float scale_modifier, const float *rotations, const float *cov3D_precomp, const int *radii_x, const int *radii_y,
    const int *radii_z, char *geom_buffer, char *binning_buffer, char *img_buffer, const float *dL_dvol, float *dL_dmean3D_norm,
    float *dL_dconic3D, float *dL_dopacity, float *dL_dmean3D, float *dL_dcov3D, float *dL_dscale, float *dL_drot,
    int debug, void *stream)
{
    (void)means3D;
    hipStream_t s = (hipStream_t)stream;
    if (P == 0) return 0;
    if (P < 0 || R < 0 || !radii_x || !radii_y || !radii_z || !geom_buffer || (R > 0 && !binning_buffer) || !dL_dvol ||
        !dL_dmean3D_norm || !dL_dconic3D || !dL_dopacity || !dL_dmean3D || !dL_dcov3D || nVoxel_x <= 0 ||
        !slab_ok(nVoxel_x, tile_x0, tile_x1) || (!cov3D_precomp && (!scales || !rotations || !dL_dscale || !dL_drot))) {
        set_error("r2_voxel_backward: invalid argument");
        return R2_ERR_INVALID;
    }
    const VoxelGrid v = make_grid(nVoxel_x, nVoxel_y, nVoxel_z, sVoxel_x, sVoxel_y, sVoxel_z, center_x, center_y, center_z, tile_x0, tile_x1);
    const VoxelGeom geom = VoxelGeom::carve(geom_buffer, P);
    const VoxelBinning bin = VoxelBinning::carve(binning_buffer, (size_t)R);
    const size_t T = (size_t)v.gx * v.gy * v.gz;
    if (R > 0 && sort_is_single_pass((int)higher_msb((uint32_t)(T > 1 ? T - 1 : 1))) && !voxel_forward_fills_tiles(T)) {
        if (!img_buffer) {
            set_error("r2_voxel_backward: image state required");
            return R2_ERR_INVALID;
        }
        const VoxelImage img = VoxelImage::carve(img_buffer, T, (size_t)v.nx * v.ny * v.nz, (size_t)R, false, vox_chunk_for(v.gy, v.gz));
        fill_tiles_from_ranges(img.ranges, T, bin.tiles, s);
    }
    // patches (the TV regulariser): nearly every gradient row is a zero row, and the zero-fill rides along with the render backward
    const bool zero_in_render = R > 0 && T <= VOX_SMALL_MAX_TILES;
    const size_t Pz = (size_t)P;
    const ZeroArrays za{{dL_dmean3D_norm, dL_dmean3D, dL_dconic3D, dL_dcov3D, dL_dopacity, dL_dscale, dL_drot},
                        {3 * Pz, 3 * Pz, 6 * Pz, 6 * Pz, Pz, dL_dscale ? 3 * Pz : 0, dL_drot ? 4 * Pz : 0}};
    { StageScope t(ST_VOX_RENDER_BWD, s);
    launch_voxel_render_backward(geom, bin, v, (size_t)R, dL_dvol, s, zero_in_render ? &za : nullptr); }
    R2_STAGE_CHECK(debug, s, "render backward");
    const float *cov3D = cov3D_precomp ? cov3D_precomp : geom.cov3D;
    { StageScope t(ST_VOX_GEOM_BWD, s);
    launch_voxel_geom_backward(geom, v, P, radii_x, radii_y, radii_z, cov3D, cov3D_precomp ? nullptr : scales,
                               cov3D_precomp ? nullptr : rotations, scale_modifier, bin.part, dL_dconic3D,
                               dL_dmean3D_norm, dL_dopacity, dL_dmean3D, dL_dcov3D, dL_dscale, dL_drot, s, zero_in_render); }
    R2_STAGE_CHECK(debug, s, "geometry backward");
    return 0;
}

extern "C" long long r2_voxel_state_offset(int which, int P, long long R, int nx, int ny, int nz, int *buffer_id)
{
    char *const base = reinterpret_cast<char *>(uintptr_t(1) << 40);
    const VoxelGrid v = make_grid(nx, ny, nz, 1, 1, 1, 0, 0, 0);
    const VoxelGeom g = VoxelGeom::carve(base, P);
    const VoxelBinning b = VoxelBinning::carve(base, (size_t)R);
    const VoxelImage im = VoxelImage::carve(base, (size_t)v.gx * v.gy * v.gz, (size_t)nx * ny * nz, (size_t)R, true, vox_chunk_for(v.gy, v.gz));
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
    case 11: p = (char *)g.first; buf = 0; break;
    case 12: p = (char *)g.order; buf = 0; break;
    case 13: p = (char *)b.inv; buf = 1; break;
    case 15: p = (char *)g.host_words; buf = 0; break;
    default: return -1;
    }
    if (buffer_id) *buffer_id = buf;
    return (long long)(p - base);
}
