"""Shared scenario builders and state readers for the parity tests."""
import ctypes as C

import numpy as np
import torch

from r2_gaussian_amd import scene as S

LOG2E = 1.4426950408889634


def tiles_from_ranges(ranges, R):
    """Tile id of every sorted instance, from the per-tile ranges (the tile-sorted list is the concatenation of the
    tile lists in tile order; empty tiles hold (0, 0))."""
    lengths = (ranges[:, 1].astype(np.int64) - ranges[:, 0].astype(np.int64))
    assert (lengths >= 0).all() and lengths.sum() == R, "ranges do not partition the sorted list"
    nz = np.nonzero(lengths)[0]
    starts = ranges[nz, 0].astype(np.int64)
    assert np.array_equal(starts, np.cumsum(lengths[nz]) - lengths[nz]), "ranges out of tile order"
    return np.repeat(np.arange(ranges.shape[0], dtype=np.uint32), lengths)


def perm_from_inv(inv):
    """The state holds inv[emission index] = sorted position; perm[sorted position] = emission index."""
    perm = np.empty_like(inv)
    assert np.array_equal(np.sort(inv), np.arange(inv.size, dtype=inv.dtype)), "inv is not a permutation"
    perm[inv] = np.arange(inv.size, dtype=inv.dtype)
    return perm


def np_view(v):
    return v.world_view_transform.numpy(), v.full_proj_transform.numpy()


def cloud_np(c):
    return c.xyz.numpy(), c.density.numpy(), c.scales.numpy(), c.rotations.numpy()


def oracle_raster(O, c, v, cov3D_precomp=None, scale_modifier=1.0, render=True):
    xyz, rho, sc, q = cloud_np(c)
    vm, pm = np_view(v)
    if cov3D_precomp is not None:
        sc, q = None, None
    return O.raster_forward(xyz, rho, sc, q, scale_modifier, cov3D_precomp, vm, pm, v.tanfovx, v.tanfovy,
                            v.image_height, v.image_width, v.mode, render=render)


def hip_raster(c, v, dev, debug=False, cov3D_precomp=None, scale_modifier=1.0, count_of=-1):
    """Run the HIP forward through the `_C` mirror and read every private intermediate back.  count_of: for a forward that returns
    a deferred-count TOKEN instead of num_rendered (r2_defer_count_control): the count to read the state back with (None: take it
    from the state's own host words); out["num_rendered"] stays what the call returned."""
    from r2_gaussian_amd import _C, _lib
    e = torch.empty(0)
    sc, q = (c.scales.to(dev), c.rotations.to(dev)) if cov3D_precomp is None else (e, e)
    cp = e if cov3D_precomp is None else torch.as_tensor(cov3D_precomp).to(dev)
    args = (c.xyz.to(dev), c.density.to(dev), sc, q, scale_modifier, cp, v.world_view_transform.to(dev),
            v.full_proj_transform.to(dev), v.tanfovx, v.tanfovy, v.image_height, v.image_width,
            v.camera_center.to(dev), False, v.mode, debug)
    R, color, radii, geom, binning, img = _C.rasterize_gaussians(*args)
    torch.cuda.synchronize()
    P, H, W = c.xyz.shape[0], v.image_height, v.image_width
    out = dict(num_rendered=R, color=color.cpu().numpy(), radii=radii.cpu().numpy(), bufs=(geom, binning, img),
               args=args)
    if P == 0:
        return out
    bufs = [geom.cpu().numpy(), binning.cpu().numpy(), img.cpu().numpy()]
    L = _lib.lib()
    if R >= 0x40000000:   # a token: the true count is in the geometry state's host words (independent of R)
        bid0 = C.c_int(-1)
        off0 = L.r2_raster_state_offset(15, P, 0, W, H, C.byref(bid0))
        true_R = int(bufs[0][off0:off0 + 4].view(np.uint32)[0])
        assert count_of is None or count_of == true_R
        R = true_R
    T = ((W + 15) // 16) * ((H + 15) // 16)

    def read(which, dtype, count):
        bid = C.c_int(-1)
        off = L.r2_raster_state_offset(which, P, R, W, H, C.byref(bid))
        assert off >= 0
        nbytes = np.dtype(dtype).itemsize * count
        return bufs[bid.value][off:off + nbytes].view(dtype).copy()

    out["tiles_touched"] = read(0, np.uint32, P)
    out["offsets"] = read(1, np.uint32, P)
    out["tiles_unsorted"] = read(2, np.uint32, R)
    out["vals_unsorted"] = read(3, np.uint32, R)
    out["point_list"] = read(5, np.uint32, R)
    out["ranges"] = read(6, np.uint32, 2 * T).reshape(T, 2)
    out["tiles"] = tiles_from_ranges(out["ranges"], R)
    out["cov3D"] = read(7, np.float32, 6 * P).reshape(P, 6)
    if debug:
        out["n_contrib"] = read(8, np.uint32, H * W)
    rec = read(9, np.float32, 8 * P).reshape(P, 8)
    out["rec"] = rec
    out["depth_key"] = read(10, np.uint32, P)
    out["depths"] = out["depth_key"].view(np.float32)
    out["first"] = read(11, np.uint32, P)
    out["order"] = read(12, np.uint32, P)
    out["host_words"] = read(15, np.uint32, 8)   # {num_rendered, overflow, thin, key extrema x4, nvis (hinted path only)}
    if debug and max(int(T) - 1, 1).bit_length() <= 12:   # inverse permutation of the single-pass tile sort: debug mode only
        out["inv"] = read(13, np.uint32, R)
        out["perm"] = perm_from_inv(out["inv"])
    # the reference's 64-bit sort keys, reconstructed: (tile << 32) | depth bits of the listed Gaussian
    out["keys"] = (out["tiles"].astype(np.uint64) << np.uint64(32)) | out["depth_key"][out["point_list"]].astype(np.uint64)
    # decode the packed render record back to the reference's quantities
    out["means2D"] = rec[:, 0:2]
    out["conic"] = np.stack([rec[:, 2] / (-0.5 * LOG2E), rec[:, 3] / (-LOG2E), rec[:, 4] / (-0.5 * LOG2E)], 1)
    om = read(14, np.float32, 2 * P).reshape(P, 2)
    out["opacity"], out["mus"] = om[:, 0], om[:, 1]
    out["extent"] = rec[:, 6:8]   # (hx, hy): half-extents of the alpha >= 1e-5 bounding box used for block culling
    return out


def hip_raster_backward(h, c, v, dL, dev):
    from r2_gaussian_amd import _C
    a = h["args"]
    geom, binning, img = h["bufs"]
    radii = torch.as_tensor(h["radii"]).to(dev)
    res = _C.rasterize_gaussians_backward(a[0], radii, a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9],
                                          torch.as_tensor(dL).to(dev), a[12], geom, h["num_rendered"], binning, img,
                                          v.mode, False)
    torch.cuda.synchronize()
    names = ["dL_dmeans2D", "dL_dopacity", "dL_dmu", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations"]
    return {n: t.cpu().numpy() for n, t in zip(names, res)}


def oracle_voxel(O, c, nVoxel, sVoxel, center, scale_modifier=1.0, render=True):
    xyz, rho, sc, q = cloud_np(c)
    return O.voxel_forward(xyz, rho, sc, q, scale_modifier, None, nVoxel, sVoxel, center, render=render)


def hip_voxel(c, nVoxel, sVoxel, center, dev, debug=False, scale_modifier=1.0, slab=None):
    """slab = (tile_x0, tile_x1): the x-slab call (r2_voxel_forward_slab) on those tile layers of the grid; the state and the
    volume read back are then the slab's ([x1 - x0, ny, nz], tiles renumbered from the slab's first layer)."""
    from r2_gaussian_amd import _C, _lib
    e = torch.empty(0)
    args = (c.xyz.to(dev), c.density.to(dev), c.scales.to(dev), c.rotations.to(dev), scale_modifier, e,
            nVoxel[0], nVoxel[1], nVoxel[2], sVoxel[0], sVoxel[1], sVoxel[2], center[0], center[1], center[2],
            False, debug)
    if slab is None:
        R, vol, rx, ry, rz, geom, binning, img = _C.voxelize_gaussians(*args)
    else:
        R, vol, rx, ry, rz, geom, binning, img = _C.voxelize_gaussians_slab(*args, int(slab[0]), int(slab[1]))
    torch.cuda.synchronize()
    P = c.xyz.shape[0]
    nx, ny, nz = nVoxel
    if slab is not None:
        nx = min(8 * int(slab[1]), nx) - 8 * int(slab[0])
        assert tuple(vol.shape) == (nx, ny, nz)
    out = dict(num_rendered=R, vol=vol.cpu().numpy(), radii_x=rx.cpu().numpy(), radii_y=ry.cpu().numpy(),
               radii_z=rz.cpu().numpy(), bufs=(geom, binning, img), args=args, radii_t=(rx, ry, rz))
    if P == 0:
        return out
    bufs = [geom.cpu().numpy(), binning.cpu().numpy(), img.cpu().numpy()]
    L = _lib.lib()
    T = ((nx + 7) // 8) * ((ny + 7) // 8) * ((nz + 7) // 8)

    def read(which, dtype, count):
        bid = C.c_int(-1)
        off = L.r2_voxel_state_offset(which, P, R, nx, ny, nz, C.byref(bid))
        assert off >= 0
        nbytes = np.dtype(dtype).itemsize * count
        return bufs[bid.value][off:off + nbytes].view(dtype).copy()

    out["tiles_touched"] = read(0, np.uint32, P)
    out["offsets"] = read(1, np.uint32, P)
    out["tiles_unsorted"] = read(2, np.uint32, R)
    out["vals_unsorted"] = read(3, np.uint32, R)
    out["point_list"] = read(5, np.uint32, R)
    out["ranges"] = read(6, np.uint32, 2 * T).reshape(T, 2)
    out["tiles"] = tiles_from_ranges(out["ranges"], R)
    out["cov3D"] = read(7, np.float32, 6 * P).reshape(P, 6)
    if debug:
        out["n_contrib"] = read(8, np.uint32, nx * ny * nz)
    out["depth_key"] = read(10, np.uint32, P)
    out["first"] = read(11, np.uint32, P)
    out["order"] = read(12, np.uint32, P)
    out["host_words"] = read(15, np.uint32, 8)   # {num_rendered, overflow, thin, key extrema x4, nvis (hinted path only)}
    if debug and max(int(T) - 1, 1).bit_length() <= 12:   # inverse permutation: single-pass (<= 12-bit) tile sort, debug mode
        out["inv"] = read(13, np.uint32, R)
        out["perm"] = perm_from_inv(out["inv"])
    out["keys"] = (out["tiles"].astype(np.uint64) << np.uint64(32)) | out["depth_key"][out["point_list"]].astype(np.uint64)
    rec = read(9, np.float32, 12 * P).reshape(P, 12)
    out["rec"] = rec
    out["means3D_norm"] = rec[:, 0:3]
    sc = np.array([-0.5 * LOG2E, -LOG2E, -LOG2E, -0.5 * LOG2E, -LOG2E, -0.5 * LOG2E], np.float32)
    out["conic"] = rec[:, 4:10] / sc
    return out


def hip_voxel_backward(h, c, nVoxel, sVoxel, center, dL, dev):
    from r2_gaussian_amd import _C
    a = h["args"]
    geom, binning, img = h["bufs"]
    rx, ry, rz = h["radii_t"]
    res = _C.voxelize_gaussians_backward(a[0], rx, ry, rz, a[2], a[3], a[4], a[5], torch.as_tensor(dL).to(dev),
                                         geom, h["num_rendered"], binning, img, nVoxel[0], nVoxel[1], nVoxel[2],
                                         sVoxel[0], sVoxel[1], sVoxel[2], center[0], center[1], center[2], False)
    torch.cuda.synchronize()
    names = ["dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations"]
    return {n: t.cpu().numpy() for n, t in zip(names, res)}


def assert_close_scaled(a, b, rtol, name, atol_frac=1e-6):
    """|a-b| <= rtol*|b| + atol_frac*max|b| elementwise; reports the worst offender."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    scale = np.abs(b).max() if b.size else 0.0
    err = np.abs(a - b)
    tol = rtol * np.abs(b) + atol_frac * scale
    bad = err > tol
    if bad.any():
        i = np.argmax(err - tol)
        raise AssertionError("%s: %d/%d out of tolerance; worst |a-b|=%.3e at %s (a=%.6e b=%.6e, scale=%.3e)" % (
            name, bad.sum(), bad.size, err.flat[i], np.unravel_index(i, a.shape), a.flat[i], b.flat[i], scale))


def ellipsoid_cloud(P, seed=0, scale_mult=1.0):
    return S.make_cloud(P, seed=seed, scale_mult=scale_mult)


TF_MARK = 0x71FE   # host word DW_PMAX of a rasterizer forward that took the tile-first path


VOX_STICKS_MARK = 0x571C


def took_sticks(h):
    return int(h["host_words"][2]) == VOX_STICKS_MARK


def took_tile_first(h):
    return int(h["host_words"][3]) == TF_MARK


def check_binning(h, o):
    """Bit-exact comparison of the binning pipeline with the oracle.  The HIP path sorts the Gaussians by depth
    first and emits instances in that order (then sorts by tile only), so its UNSORTED arrays are a per-Gaussian
    permutation of the reference's; everything the reference defines -- per-Gaussian tile runs, the sorted
    (tile|depth) key list, point_list, ranges -- must match exactly."""
    P = o["P"]
    assert h["num_rendered"] == o["num_rendered"]
    R = o["num_rendered"]
    assert np.array_equal(h["tiles_touched"], o["tiles_touched"])
    tt = o["tiles_touched"].astype(np.int64)
    sticks = int(h["host_words"][2]) == VOX_STICKS_MARK
    if int(h["host_words"][2]) == 0x5A11 or int(h["host_words"][3]) == TF_MARK or sticks:
        # rasterizer, tile-first binning (csrc/raster_tilefirst.hip, marker in host word 3), or
        # voxelizer, small-grid path (csrc/voxel_small.hip): no global depth order and no emission list exist -- the per-tile
        # lists are built straight from the survivors.  What the reference defines must still match bit for bit: the sorted
        # (tile | depth) keys, point_list, ranges; and every visible Gaussian owns a run of tiles_touched scratch rows, the runs
        # disjoint and covering [0, R) (the backward's contract).
        nvis = int((tt > 0).sum())
        if sticks:   # voxelizer, stick-first binning (csrc/voxel_sticks.hip): `order` lists every id, the visible count is cleared
            assert int(h["host_words"][7]) == 0 and np.array_equal(h["order"], np.arange(P, dtype=np.uint32))
        else:
            assert int(h["host_words"][7]) == nvis
        vis = np.nonzero(tt > 0)[0]
        start = h["first"].astype(np.int64)[vis]
        srt = np.argsort(start, kind="stable")
        assert np.array_equal(np.cumsum(tt[vis][srt]) - tt[vis][srt], start[srt]), "scratch-row runs are not a partition of [0, R)"
        assert np.array_equal(h["keys"], o["keys"]), "sorted (tile|depth) keys differ"
        assert np.array_equal(h["point_list"], o["point_list"]), "point_list differs"
        assert np.array_equal(h["ranges"], o["ranges"]), "ranges differ"
        return
    # depth order: stable argsort of the depth bits; culled Gaussians emit nothing so only visible order matters
    vis = tt > 0
    # (with a depth hint only the visible prefix of order / offsets is written -- the host words then carry nvis;
    # without one `order` is a full permutation, Gaussians that emit nothing sit wherever their key puts them)
    nvis = int(vis.sum())
    order = h["order"].astype(np.int64)
    dk = o["depths"].view(np.uint32)
    ref_ov = np.nonzero(vis)[0][np.argsort(dk[vis], kind="stable")]
    if int(h["host_words"][7]) != 0 or nvis == 0:
        assert int(h["host_words"][7]) == nvis
        ov = order[:nvis]
        assert np.array_equal(ov, ref_ov), "depth order differs (hinted path)"
        assert np.array_equal(h["offsets"][:nvis].astype(np.int64), np.cumsum(tt[ov]))
    else:
        assert np.array_equal(np.sort(order), np.arange(P))
        ov = order[vis[order]]
        assert np.array_equal(ov, ref_ov), "depth order differs"
        assert np.array_equal(h["offsets"].astype(np.int64), np.cumsum(tt[order]))
    # per-Gaussian runs: same tiles in the same (y-major / x-minor) order as the reference's emission
    o_start = o["offsets"].astype(np.int64) - tt
    h_start = h["first"].astype(np.int64)
    ids = np.repeat(np.arange(P), tt)
    within = np.arange(R) - np.repeat(np.cumsum(tt) - tt, tt)
    o_tiles = (o["keys_unsorted"] >> np.uint64(32)).astype(np.uint32)[o_start[ids] + within]
    h_idx = h_start[ids] + within
    assert np.array_equal(h["tiles_unsorted"][h_idx], o_tiles), "emitted tile runs differ"
    assert np.array_equal(h["vals_unsorted"][h_idx], ids.astype(np.uint32))
    # the sorted list is the reference's, bit for bit
    assert np.array_equal(h["keys"], o["keys"]), "sorted (tile|depth) keys differ"
    assert np.array_equal(h["point_list"], o["point_list"]), "point_list differs"
    assert np.array_equal(h["ranges"], o["ranges"]), "ranges differ"
    if "perm" in h:
        assert np.array_equal(h["vals_unsorted"][h["perm"]], h["point_list"])


# ------------------------------------------------------------------------------------------------ value parity
# north_star: "rendered projections and voxel grids within 1e-4 rel".  The checks below assert exactly that -- the pure
# relative bound, no absolute floor -- and attribute every excess to a counted cut-off flip (oracle/parity.py).  Every
# check's statistics are collected in PARITY_LOG and written to gpurun_out/parity_report.json at session end
# (tests/conftest.py), so the margins are visible even when everything passes.
PARITY_LOG = []


def _log(kind, label, fn):
    """Run a parity check; its statistics (or its failure text) go to PARITY_LOG either way."""
    from oracle import parity as Pz
    try:
        stats = fn()
    except Pz.ParityError as e:
        PARITY_LOG.append(dict(kind=kind, case=label, error=str(e)[:4000]))
        raise
    PARITY_LOG.append(dict(kind=kind, case=label, **{k: v for k, v in stats.items() if not k.startswith("_")}))
    return stats


def parity_image(O, o, got, label=""):
    from oracle import parity as Pz
    budget, _nb = O.raster_forward_audit(o)
    return _log("image", label, lambda: Pz.image_parity(got, o["color"], budget, what="image " + label))


def parity_volume(O, o, got, label=""):
    from oracle import parity as Pz
    budget, _nb = O.voxel_forward_audit(o)
    return _log("volume", label, lambda: Pz.image_parity(got, o["vol"], budget, what="volume " + label))


def parity_raster_grads(O, o, gh, c, v, dL, label="", cov3D_precomp=None, scale_modifier=1.0):
    from oracle import parity as Pz
    xyz, _rho, sc, q = cloud_np(c)
    if cov3D_precomp is not None:
        sc, q = None, None
    vm, pm = np_view(v)
    return _log("raster_grads", label, lambda: Pz.raster_grad_parity(O, o, dL, gh, xyz, sc, q, scale_modifier, cov3D_precomp,
                                                                     vm, pm, v.tanfovx, v.tanfovy))


def parity_voxel_grads(O, o, gh, c, dL, label="", scale_modifier=1.0):
    from oracle import parity as Pz
    return _log("voxel_grads", label, lambda: Pz.voxel_grad_parity(O, o, dL, gh, c.scales.numpy(), c.rotations.numpy(),
                                                                   scale_modifier, None))
