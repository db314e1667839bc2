"""numpy/ctypes front-end of the CPU parity oracle (oracle/r2_oracle.c).

TEST INFRASTRUCTURE ONLY.  Nothing under ``r2_gaussian_amd/`` may import this module; it is
used by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
as the *checker*, never as the thing measured or shipped.

The pipeline drivers below restate the host-side orchestration of
``CudaRasterizer::Rasterizer::forward/backward`` (SUB/cuda_rasterizer/rasterizer_impl.cu:196-421)
and ``CudaVoxelizer::Voxelizer::forward/backward`` (SUB/cuda_voxelizer/voxelizer_impl.cu:171-389),
plus the zero-initialisation conventions of the torch boundary
(SUB/rasterize_points.cu:58-59,124-131, SUB/voxelize_points.cu:58-61,130-136), and expose every
intermediate (radii, tiles_touched, offsets, keys, sorted point list, ranges, n_contrib) because
tile / sort indices must match bit-exactly.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f = np.float32


def build(force=False):
    """Compile liboracle with gcc (seconds)."""
    so = os.path.join(_HERE, "libr2oracle.so")
    src = os.path.join(_HERE, "r2_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libr2oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.r2o_higher_msb.restype = C.c_uint32
        _LIB.r2o_higher_msb.argtypes = [C.c_uint32]
        _LIB.r2o_inclusive_scan.restype = C.c_uint32
    return _LIB


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _c32(x):
    return None if x is None else np.ascontiguousarray(np.asarray(x, dtype=_f))


def _empty_to_none(x):
    if x is None:
        return None
    x = np.asarray(x)
    return None if x.size == 0 else x


def higher_msb(n):
    return int(lib().r2o_higher_msb(int(n)))


def cov3d(scales, scale_modifier, rotations):
    scales, rotations = _c32(scales), _c32(rotations)
    P = scales.shape[0]
    out = np.zeros((P, 6), _f)
    lib().r2o_cov3d(C.c_int(P), _p(scales), C.c_float(scale_modifier), _p(rotations), _p(out))
    return out


def sort_pairs(keys, vals, end_bit):
    R = keys.shape[0]
    ko, vo = np.empty(R, np.uint64), np.empty(R, np.uint32)
    lib().r2o_sort_pairs(C.c_int64(R), _p(keys), _p(vals), _p(ko), _p(vo), C.c_int(end_bit))
    return ko, vo


def mark_visible(means3D, viewmatrix, projmatrix):
    means3D = _c32(means3D)
    P = means3D.shape[0]
    out = np.zeros(P, np.uint8)
    lib().r2o_mark_visible(C.c_int(P), _p(means3D), _p(_c32(viewmatrix).reshape(-1)), _p(_c32(projmatrix).reshape(-1)), _p(out))
    return out.astype(bool)


# --------------------------------------------------------------------------- rasterizer
def raster_forward(means3D, opacities, scales, rotations, scale_modifier, cov3D_precomp,
                   viewmatrix, projmatrix, tanfovx, tanfovy, image_height, image_width, mode,
                   render=True):
    """Restates Rasterizer::forward.  Matrices are the flat 16-float arrays the kernels index
    column-major (i.e. the row-major memory of ``world_view_transform`` / ``full_proj_transform``)."""
    L = lib()
    means3D = _c32(means3D).reshape(-1, 3)
    P = means3D.shape[0]
    H, W = int(image_height), int(image_width)
    opacities = _c32(opacities).reshape(-1)
    scales = _c32(_empty_to_none(scales))
    rotations = _c32(_empty_to_none(rotations))
    cov3D_precomp = _c32(_empty_to_none(cov3D_precomp))
    view = _c32(viewmatrix).reshape(-1)
    proj = _c32(projmatrix).reshape(-1)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T = gx * gy
    st = dict(P=P, H=H, W=W, grid=(gx, gy), mode=int(mode))
    st["radii"] = np.zeros(P, np.int32)
    st["means2D"] = np.zeros((P, 2), _f)
    st["depths"] = np.zeros(P, _f)
    st["cov3D"] = np.zeros((P, 6), _f)
    st["conic_opacity"] = np.zeros((P, 4), _f)
    st["mus"] = np.zeros(P, _f)
    st["tiles_touched"] = np.zeros(P, np.uint32)
    st["color"] = np.zeros((1, H, W), _f)
    st["n_contrib"] = np.zeros(H * W, np.uint32)
    st["ranges"] = np.zeros((T, 2), np.uint32)
    st["num_rendered"] = 0
    if P == 0:
        st["offsets"] = np.zeros(0, np.uint32)
        st["keys_unsorted"] = st["keys"] = np.zeros(0, np.uint64)
        st["vals_unsorted"] = st["point_list"] = np.zeros(0, np.uint32)
        return st
    L.r2o_raster_preprocess(
        C.c_int(P), _p(means3D), _p(scales), C.c_float(scale_modifier), _p(rotations), _p(opacities),
        _p(cov3D_precomp), _p(view), _p(proj), C.c_int(W), C.c_int(H), C.c_float(tanfovx), C.c_float(tanfovy),
        C.c_int(mode), _p(st["radii"]), _p(st["means2D"]), _p(st["depths"]), _p(st["cov3D"]),
        _p(st["conic_opacity"]), _p(st["mus"]), _p(st["tiles_touched"]))
    st["offsets"] = np.zeros(P, np.uint32)
    R = int(L.r2o_inclusive_scan(C.c_int(P), _p(st["tiles_touched"]), _p(st["offsets"])))
    st["num_rendered"] = R
    ku, vu = np.zeros(R, np.uint64), np.zeros(R, np.uint32)
    L.r2o_raster_duplicate(C.c_int(P), _p(st["means2D"]), _p(st["depths"]), _p(st["offsets"]), _p(st["radii"]),
                           C.c_int(gx), C.c_int(gy), _p(ku), _p(vu))
    st["keys_unsorted"], st["vals_unsorted"] = ku, vu
    st["sort_bits"] = 32 + higher_msb(T)
    st["keys"], st["point_list"] = sort_pairs(ku, vu, st["sort_bits"])
    L.r2o_tile_ranges(C.c_int64(R), _p(st["keys"]), C.c_int64(T), _p(st["ranges"]))
    if render:
        L.r2o_raster_render_fwd(_p(st["ranges"]), _p(st["point_list"]), C.c_int(W), C.c_int(H), _p(st["means2D"]),
                                _p(st["conic_opacity"]), _p(st["mus"]), _p(st["n_contrib"]), _p(st["color"]))
    return st


def raster_backward(st, means3D, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
                    tanfovx, tanfovy, dL_dcolor, acc64=False):
    """Restates RasterizeGaussiansBackwardCUDA + Rasterizer::backward: render -> computeCov2D -> preprocess."""
    L = lib()
    P, H, W, mode = st["P"], st["H"], st["W"], st["mode"]
    means3D = _c32(means3D).reshape(-1, 3)
    scales = _c32(_empty_to_none(scales))
    rotations = _c32(_empty_to_none(rotations))
    cov3D_precomp = _c32(_empty_to_none(cov3D_precomp))
    view = _c32(viewmatrix).reshape(-1)
    proj = _c32(projmatrix).reshape(-1)
    dL_dcolor = _c32(dL_dcolor).reshape(-1)
    g = dict(
        dL_dmeans3D=np.zeros((P, 3), _f), dL_dmeans2D=np.zeros((P, 3), _f), dL_dconic=np.zeros((P, 2, 2), _f),
        dL_dopacity=np.zeros((P, 1), _f), dL_dmu=np.zeros((P, 1), _f), dL_dcov3D=np.zeros((P, 6), _f),
        dL_dscales=np.zeros((P, 3), _f), dL_drotations=np.zeros((P, 4), _f))
    if P == 0:
        return g
    L.r2o_raster_render_bwd(_p(st["ranges"]), _p(st["point_list"]), C.c_int(W), C.c_int(H), C.c_int(P),
                            _p(st["means2D"]), _p(st["conic_opacity"]), _p(st["mus"]), _p(st["n_contrib"]),
                            _p(dL_dcolor), _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dopacity"]),
                            _p(g["dL_dmu"]), C.c_int(1 if acc64 else 0))
    cov3D = cov3D_precomp if cov3D_precomp is not None else st["cov3D"]
    L.r2o_raster_cov2d_bwd(C.c_int(P), _p(means3D), _p(st["radii"]), _p(cov3D), C.c_int(W), C.c_int(H),
                           C.c_float(tanfovx), C.c_float(tanfovy), _p(view), _p(g["dL_dconic"]), _p(g["dL_dmu"]),
                           _p(g["dL_dmeans3D"]), _p(g["dL_dcov3D"]), C.c_int(mode))
    L.r2o_raster_preprocess_bwd(C.c_int(P), _p(means3D), _p(st["radii"]), _p(scales), _p(rotations),
                                C.c_float(scale_modifier), _p(proj), _p(g["dL_dmeans2D"]), _p(g["dL_dmeans3D"]),
                                _p(g["dL_dcov3D"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]))
    return g


def raster_forward_audit(st):
    """Per pixel: (flip_budget, n_border) -- how much of the pixel hangs on pairs that sit ON one of the reference's cut-off
    tests (see the AUDIT section of r2_oracle.c).  A parity test asserts |got - ref| <= rtol*|ref| + flip_budget."""
    H, W = st["H"], st["W"]
    budget, nb = np.zeros((1, H, W), _f), np.zeros(H * W, np.uint32)
    if st["P"]:
        lib().r2o_raster_render_fwd_audit(_p(st["ranges"]), _p(st["point_list"]), C.c_int(W), C.c_int(H), _p(st["means2D"]),
                                          _p(st["conic_opacity"]), _p(st["mus"]), _p(budget), _p(nb))
    return budget, nb.reshape(H, W)


RASTER_RAW = ("mean2D_x", "mean2D_y", "conic_x", "conic_y", "conic_w", "opacity", "mu")


PAIR_CAP = 1 << 20   # borderline pairs listed per call (a case with more is reported as truncated, never silently cut)


def _pair_buffers(width, pairs):
    cap = PAIR_CAP if pairs else 0
    return cap, np.zeros(max(cap, 1), np.uint32), np.zeros((max(cap, 1), width), np.float64), C.c_int64(0)


def _pair_result(cap, ids, vals, count):
    n = int(count.value)
    return {"ids": ids[:min(n, cap)], "vals": vals[:min(n, cap)], "truncated": n > cap, "count": n}


def raster_backward_audit(st, dL_dcolor, pairs=False):
    """The 7 raw sums of the render backward per Gaussian, accumulated in double: (sum, abssum, flip), each [P,7] in the order
    RASTER_RAW.  abssum = sum of |terms| (the scale a float sum with cancellation is accurate to), flip = |terms| of pairs on a
    cut-off.  pairs=True: a fourth result, the list of the borderline pairs themselves -- dict(ids [n] Gaussian, vals [n,7] what
    the pair would ADD to that Gaussian's sums if the cut-off were decided the other way)."""
    P, H, W = st["P"], st["H"], st["W"]
    out = [np.zeros((P, 7), np.float64) for _ in range(3)]
    cap, ids, vals, count = _pair_buffers(7, pairs)
    if P:
        lib().r2o_raster_render_bwd_audit(_p(st["ranges"]), _p(st["point_list"]), C.c_int(W), C.c_int(H), C.c_int(P),
                                          C.c_int64(st["num_rendered"]), _p(st["means2D"]), _p(st["conic_opacity"]),
                                          _p(st["mus"]), _p(st["n_contrib"]), _p(_c32(dL_dcolor).reshape(-1)),
                                          _p(out[0]), _p(out[1]), _p(out[2]), C.c_int64(cap), _p(ids), _p(vals), C.byref(count))
    return out + [_pair_result(cap, ids, vals, count)] if pairs else out


def raster_geom_chain(st, raw, means3D, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tanfovx,
                      tanfovy):
    """computeCov2DCUDA + preprocessCUDA backward applied to given raw sums ([P,7], RASTER_RAW order): the (float-evaluated)
    linear map from the render backward's sums to the returned gradients."""
    L = lib()
    P, H, W, mode = st["P"], st["H"], st["W"], st["mode"]
    raw = np.asarray(raw)
    g = dict(dL_dmeans3D=np.zeros((P, 3), _f), dL_dmeans2D=np.zeros((P, 3), _f), dL_dconic=np.zeros((P, 2, 2), _f),
             dL_dopacity=np.zeros((P, 1), _f), dL_dmu=np.zeros((P, 1), _f), dL_dcov3D=np.zeros((P, 6), _f),
             dL_dscales=np.zeros((P, 3), _f), dL_drotations=np.zeros((P, 4), _f))
    g["dL_dmeans2D"][:, 0:2] = raw[:, 0:2]
    g["dL_dconic"].reshape(P, 4)[:, [0, 1, 3]] = raw[:, 2:5]
    g["dL_dopacity"][:, 0] = raw[:, 5]
    g["dL_dmu"][:, 0] = raw[:, 6]
    means3D = _c32(means3D).reshape(-1, 3)
    scales = _c32(_empty_to_none(scales))
    rotations = _c32(_empty_to_none(rotations))
    cov3D_precomp = _c32(_empty_to_none(cov3D_precomp))
    view, proj = _c32(viewmatrix).reshape(-1), _c32(projmatrix).reshape(-1)
    cov3D = cov3D_precomp if cov3D_precomp is not None else st["cov3D"]
    L.r2o_raster_cov2d_bwd(C.c_int(P), _p(means3D), _p(st["radii"]), _p(cov3D), C.c_int(W), C.c_int(H),
                           C.c_float(tanfovx), C.c_float(tanfovy), _p(view), _p(g["dL_dconic"]), _p(g["dL_dmu"]),
                           _p(g["dL_dmeans3D"]), _p(g["dL_dcov3D"]), C.c_int(mode))
    L.r2o_raster_preprocess_bwd(C.c_int(P), _p(means3D), _p(st["radii"]), _p(scales), _p(rotations),
                                C.c_float(scale_modifier), _p(proj), _p(g["dL_dmeans2D"]), _p(g["dL_dmeans3D"]),
                                _p(g["dL_dcov3D"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]))
    return g


# --------------------------------------------------------------------------- voxelizer
def voxel_forward(means3D, opacities, scales, rotations, scale_modifier, cov3D_precomp,
                  nVoxel, sVoxel, center, render=True):
    L = lib()
    means3D = _c32(means3D).reshape(-1, 3)
    P = means3D.shape[0]
    nx, ny, nz = (int(v) for v in nVoxel)
    sx, sy, sz = (float(v) for v in sVoxel)
    cx, cy, cz = (float(v) for v in center)
    opacities = _c32(opacities).reshape(-1)
    scales = _c32(_empty_to_none(scales))
    rotations = _c32(_empty_to_none(rotations))
    cov3D_precomp = _c32(_empty_to_none(cov3D_precomp))
    gx, gy, gz = (nx + 7) // 8, (ny + 7) // 8, (nz + 7) // 8
    T = gx * gy * gz
    st = dict(P=P, nVoxel=(nx, ny, nz), sVoxel=(sx, sy, sz), center=(cx, cy, cz), grid=(gx, gy, gz))
    for k in ("radii_x", "radii_y", "radii_z"):
        st[k] = np.zeros(P, np.int32)
    st["means3D_norm"] = np.zeros((P, 3), _f)
    st["depths"] = np.zeros(P, _f)
    st["cov3D"] = np.zeros((P, 6), _f)
    st["conic_opacity"] = np.zeros((P, 7), _f)
    st["tiles_touched"] = np.zeros(P, np.uint32)
    st["vol"] = np.zeros((nx, ny, nz), _f)
    st["n_contrib"] = np.zeros(nx * ny * nz, np.uint32)
    st["ranges"] = np.zeros((T, 2), np.uint32)
    st["num_rendered"] = 0
    if P == 0:
        st["offsets"] = np.zeros(0, np.uint32)
        st["keys_unsorted"] = st["keys"] = np.zeros(0, np.uint64)
        st["vals_unsorted"] = st["point_list"] = np.zeros(0, np.uint32)
        return st
    L.r2o_voxel_preprocess(
        C.c_int(P), _p(means3D), _p(scales), C.c_float(scale_modifier), _p(rotations), _p(opacities), _p(cov3D_precomp),
        C.c_int(nx), C.c_int(ny), C.c_int(nz), C.c_float(sx), C.c_float(sy), C.c_float(sz),
        C.c_float(cx), C.c_float(cy), C.c_float(cz), _p(st["radii_x"]), _p(st["radii_y"]), _p(st["radii_z"]),
        _p(st["means3D_norm"]), _p(st["depths"]), _p(st["cov3D"]), _p(st["conic_opacity"]), _p(st["tiles_touched"]))
    st["offsets"] = np.zeros(P, np.uint32)
    R = int(L.r2o_inclusive_scan(C.c_int(P), _p(st["tiles_touched"]), _p(st["offsets"])))
    st["num_rendered"] = R
    ku, vu = np.zeros(R, np.uint64), np.zeros(R, np.uint32)
    L.r2o_voxel_duplicate(C.c_int(P), _p(st["means3D_norm"]), _p(st["depths"]), _p(st["offsets"]),
                          _p(st["radii_x"]), _p(st["radii_y"]), _p(st["radii_z"]),
                          C.c_int(gx), C.c_int(gy), C.c_int(gz), _p(ku), _p(vu))
    st["keys_unsorted"], st["vals_unsorted"] = ku, vu
    st["sort_bits"] = 32 + higher_msb(T)
    st["keys"], st["point_list"] = sort_pairs(ku, vu, st["sort_bits"])
    L.r2o_tile_ranges(C.c_int64(R), _p(st["keys"]), C.c_int64(T), _p(st["ranges"]))
    if render:
        L.r2o_voxel_render_fwd(_p(st["ranges"]), _p(st["point_list"]), C.c_int(nx), C.c_int(ny), C.c_int(nz),
                               _p(st["means3D_norm"]), _p(st["conic_opacity"]), _p(st["n_contrib"]), _p(st["vol"]))
    return st


def voxel_backward(st, scales, rotations, scale_modifier, cov3D_precomp, dL_dvol, acc64=False):
    L = lib()
    P = st["P"]
    nx, ny, nz = st["nVoxel"]
    sx, sy, sz = st["sVoxel"]
    scales = _c32(_empty_to_none(scales))
    rotations = _c32(_empty_to_none(rotations))
    cov3D_precomp = _c32(_empty_to_none(cov3D_precomp))
    dL_dvol = _c32(dL_dvol).reshape(-1)
    g = dict(
        dL_dmeans3D=np.zeros((P, 3), _f), dL_dmeans3D_norm=np.zeros((P, 3), _f), dL_dconic3D=np.zeros((P, 6), _f),
        dL_dopacity=np.zeros((P, 1), _f), dL_dcov3D=np.zeros((P, 6), _f), dL_dscales=np.zeros((P, 3), _f),
        dL_drotations=np.zeros((P, 4), _f))
    if P == 0:
        return g
    L.r2o_voxel_render_bwd(_p(st["ranges"]), _p(st["point_list"]), C.c_int(nx), C.c_int(ny), C.c_int(nz),
                           C.c_float(sx), C.c_float(sy), C.c_float(sz), C.c_int(P), _p(st["means3D_norm"]),
                           _p(st["conic_opacity"]), _p(st["n_contrib"]), _p(dL_dvol), _p(g["dL_dmeans3D_norm"]),
                           _p(g["dL_dconic3D"]), _p(g["dL_dopacity"]), C.c_int(1 if acc64 else 0))
    cov3D = cov3D_precomp if cov3D_precomp is not None else st["cov3D"]
    L.r2o_voxel_cov3d_bwd(C.c_int(P), _p(st["radii_x"]), _p(st["radii_y"]), _p(st["radii_z"]), _p(cov3D),
                          C.c_int(nx), C.c_int(ny), C.c_int(nz), C.c_float(sx), C.c_float(sy), C.c_float(sz),
                          _p(g["dL_dconic3D"]), _p(g["dL_dcov3D"]))
    L.r2o_voxel_preprocess_bwd(C.c_int(P), _p(st["radii_x"]), _p(st["radii_y"]), _p(st["radii_z"]), _p(scales),
                               _p(rotations), C.c_float(scale_modifier), _p(g["dL_dmeans3D_norm"]),
                               _p(g["dL_dmeans3D"]), _p(g["dL_dcov3D"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]))
    return g


def voxel_forward_audit(st):
    nx, ny, nz = st["nVoxel"]
    budget, nb = np.zeros((nx, ny, nz), _f), np.zeros(nx * ny * nz, np.uint32)
    if st["P"]:
        lib().r2o_voxel_render_fwd_audit(_p(st["ranges"]), _p(st["point_list"]), C.c_int(nx), C.c_int(ny), C.c_int(nz),
                                         _p(st["means3D_norm"]), _p(st["conic_opacity"]), _p(budget), _p(nb))
    return budget, nb.reshape(nx, ny, nz)


VOXEL_RAW = ("mean_x", "mean_y", "mean_z", "conic_0", "conic_1", "conic_2", "conic_3", "conic_4", "conic_5", "opacity")


def voxel_backward_audit(st, dL_dvol, pairs=False):
    """(sum, abssum, flip), each [P,10] float64 in VOXEL_RAW order (+ the pair list with pairs=True) -- see
    raster_backward_audit."""
    P = st["P"]
    nx, ny, nz = st["nVoxel"]
    sx, sy, sz = st["sVoxel"]
    out = [np.zeros((P, 10), np.float64) for _ in range(3)]
    cap, ids, vals, count = _pair_buffers(10, pairs)
    if P:
        lib().r2o_voxel_render_bwd_audit(_p(st["ranges"]), _p(st["point_list"]), C.c_int(nx), C.c_int(ny), C.c_int(nz),
                                         C.c_float(sx), C.c_float(sy), C.c_float(sz), C.c_int(P),
                                         C.c_int64(st["num_rendered"]), _p(st["means3D_norm"]), _p(st["conic_opacity"]),
                                         _p(st["n_contrib"]), _p(_c32(dL_dvol).reshape(-1)), _p(out[0]), _p(out[1]), _p(out[2]),
                                         C.c_int64(cap), _p(ids), _p(vals), C.byref(count))
    return out + [_pair_result(cap, ids, vals, count)] if pairs else out


def voxel_geom_chain(st, raw, scales, rotations, scale_modifier, cov3D_precomp):
    """computeCov3DCUDA + preprocessCUDA backward (VOX/backward.cu:86-213) applied to given raw sums ([P,10], VOXEL_RAW)."""
    L = lib()
    P = st["P"]
    nx, ny, nz = st["nVoxel"]
    sx, sy, sz = st["sVoxel"]
    raw = np.asarray(raw)
    scales = _c32(_empty_to_none(scales))
    rotations = _c32(_empty_to_none(rotations))
    cov3D_precomp = _c32(_empty_to_none(cov3D_precomp))
    g = dict(dL_dmeans3D=np.zeros((P, 3), _f), dL_dmeans3D_norm=np.ascontiguousarray(raw[:, 0:3], _f),
             dL_dconic3D=np.ascontiguousarray(raw[:, 3:9], _f), dL_dopacity=np.ascontiguousarray(raw[:, 9:10], _f),
             dL_dcov3D=np.zeros((P, 6), _f), dL_dscales=np.zeros((P, 3), _f), dL_drotations=np.zeros((P, 4), _f))
    cov3D = cov3D_precomp if cov3D_precomp is not None else st["cov3D"]
    L.r2o_voxel_cov3d_bwd(C.c_int(P), _p(st["radii_x"]), _p(st["radii_y"]), _p(st["radii_z"]), _p(cov3D),
                          C.c_int(nx), C.c_int(ny), C.c_int(nz), C.c_float(sx), C.c_float(sy), C.c_float(sz),
                          _p(g["dL_dconic3D"]), _p(g["dL_dcov3D"]))
    L.r2o_voxel_preprocess_bwd(C.c_int(P), _p(st["radii_x"]), _p(st["radii_y"]), _p(st["radii_z"]), _p(scales),
                               _p(rotations), C.c_float(scale_modifier), _p(g["dL_dmeans3D_norm"]),
                               _p(g["dL_dmeans3D"]), _p(g["dL_dcov3D"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]))
    return g


def knn_dist2(points):
    points = _c32(points).reshape(-1, 3)
    P = points.shape[0]
    out = np.zeros(P, _f)
    if P:
        lib().r2o_knn_dist2(C.c_int(P), _p(points), _p(out))
    return out
