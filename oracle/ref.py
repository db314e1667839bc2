"""ctypes front-end of oracle/_ref: the REFERENCE's own CUDA kernels (SUB/cuda_rasterizer/*.cu,
SUB/cuda_voxelizer/*.cu) compiled for the host CPU by ``make -C oracle ref`` (see oracle/Makefile for what is
and is not the reference's code in that build).

TEST INFRASTRUCTURE ONLY.  It exists to PIN the hand-written oracle (oracle/r2_oracle.c): in this container
(where /root/reference exists) the two are compared bit for bit, and tests/golden/*.npz are generated from it
(tests/golden/make_golden.py) so that the pinning travels to machines without the reference tree.

Returns the same dict layout as oracle.raster_forward / raster_backward / voxel_forward / voxel_backward.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = "/root/reference/r2_gaussian/submodules/xray-gaussian-rasterization-voxelization"
_RAS = _VOX = None
_f = np.float32


def available():
    """True when the prebuilt _ref libraries exist or can be built (the reference tree is present)."""
    return (os.path.exists(os.path.join(_HERE, "_ref", "libr2ref_raster.so")) and
            os.path.exists(os.path.join(_HERE, "_ref", "libr2ref_voxel.so"))) or os.path.isdir(REF_ROOT)


def build():
    """(Re)build oracle/_ref from the reference sources where they lie; a no-op without /root/reference."""
    if os.path.isdir(REF_ROOT):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
        return True
    return False


def _libs():
    global _RAS, _VOX
    if _RAS is None:
        build()
        _RAS = C.CDLL(os.path.join(_HERE, "_ref", "libr2ref_raster.so"))
        _VOX = C.CDLL(os.path.join(_HERE, "_ref", "libr2ref_voxel.so"))
    return _RAS, _VOX


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _c32(x, shape=None):
    if x is None:
        return None
    x = np.asarray(x)
    if x.size == 0:
        return None
    x = np.ascontiguousarray(x, dtype=_f)
    return x.reshape(shape) if shape else x


def raster_forward(means3D, opacities, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
                   tanfovx, tanfovy, image_height, image_width, mode):
    L, _ = _libs()
    means3D = np.ascontiguousarray(np.asarray(means3D, _f).reshape(-1, 3))
    P, H, W = means3D.shape[0], int(image_height), int(image_width)
    opac = np.ascontiguousarray(np.asarray(opacities, _f).reshape(-1))
    scales, rotations, cov = _c32(scales), _c32(rotations), _c32(cov3D_precomp)
    view, proj = _c32(viewmatrix, (-1,)), _c32(projmatrix, (-1,))
    campos = np.zeros(3, _f)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    st = dict(P=P, H=H, W=W, grid=(gx, gy), mode=int(mode))
    st["color"] = np.zeros((1, H, W), _f)
    st["radii"] = np.zeros(P, np.int32)
    R = L.r2ref_raster_forward(C.c_int(P), C.c_int(W), C.c_int(H), _p(means3D), _p(opac), _p(scales),
                               C.c_float(scale_modifier), _p(rotations), _p(cov), _p(view), _p(proj), _p(campos),
                               C.c_float(tanfovx), C.c_float(tanfovy), C.c_int(mode), _p(st["color"]), _p(st["radii"]))
    st["num_rendered"] = int(R)
    spec = [("depths", 0, (P,), _f), ("means2D", 1, (P, 2), _f), ("cov3D", 2, (P, 6), _f),
            ("conic_opacity", 3, (P, 4), _f), ("mus", 4, (P,), _f), ("tiles_touched", 5, (P,), np.uint32),
            ("offsets", 6, (P,), np.uint32), ("keys_unsorted", 7, (R,), np.uint64), ("vals_unsorted", 8, (R,), np.uint32),
            ("keys", 9, (R,), np.uint64), ("point_list", 10, (R,), np.uint32), ("ranges", 11, (gx * gy, 2), np.uint32),
            ("n_contrib", 12, (H * W,), np.uint32)]
    for name, which, shape, dt in spec:
        a = np.zeros(shape, dt)
        if P:
            L.r2ref_raster_get(C.c_int(which), _p(a))
        st[name] = a
    return st


def raster_backward(st, means3D, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
                    tanfovx, tanfovy, dL_dcolor):
    """Must follow the raster_forward call it belongs to (the state buffers live inside the library)."""
    L, _ = _libs()
    P = st["P"]
    means3D = np.ascontiguousarray(np.asarray(means3D, _f).reshape(-1, 3))
    scales, rotations, cov = _c32(scales), _c32(rotations), _c32(cov3D_precomp)
    view, proj = _c32(viewmatrix, (-1,)), _c32(projmatrix, (-1,))
    dL = _c32(dL_dcolor, (-1,))
    campos = np.zeros(3, _f)
    g = dict(dL_dmeans3D=np.zeros((P, 3), _f), dL_dmeans2D=np.zeros((P, 3), _f), dL_dconic=np.zeros((P, 2, 2), _f),
             dL_dopacity=np.zeros((P, 1), _f), dL_dmu=np.zeros((P, 1), _f), dL_dcov3D=np.zeros((P, 6), _f),
             dL_dscales=np.zeros((P, 3), _f), dL_drotations=np.zeros((P, 4), _f))
    L.r2ref_raster_backward(_p(means3D), _p(scales), C.c_float(scale_modifier), _p(rotations), _p(cov), _p(view), _p(proj),
                            _p(campos), C.c_float(tanfovx), C.c_float(tanfovy), _p(st["radii"]), C.c_int(st["mode"]),
                            _p(dL), _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dopacity"]), _p(g["dL_dmu"]),
                            _p(g["dL_dmeans3D"]), _p(g["dL_dcov3D"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]))
    return g


def mark_visible(means3D, viewmatrix, projmatrix):
    L, _ = _libs()
    means3D = np.ascontiguousarray(np.asarray(means3D, _f).reshape(-1, 3))
    out = np.zeros(means3D.shape[0], np.uint8)
    L.r2ref_mark_visible(C.c_int(means3D.shape[0]), _p(means3D), _p(_c32(viewmatrix, (-1,))), _p(_c32(projmatrix, (-1,))),
                         _p(out))
    return out.astype(bool)


def voxel_forward(means3D, opacities, scales, rotations, scale_modifier, cov3D_precomp, nVoxel, sVoxel, center):
    _, L = _libs()
    means3D = np.ascontiguousarray(np.asarray(means3D, _f).reshape(-1, 3))
    P = means3D.shape[0]
    nx, ny, nz = (int(v) for v in nVoxel)
    sx, sy, sz = (float(v) for v in sVoxel)
    cx, cy, cz = (float(v) for v in center)
    opac = np.ascontiguousarray(np.asarray(opacities, _f).reshape(-1))
    scales, rotations, cov = _c32(scales), _c32(rotations), _c32(cov3D_precomp)
    gx, gy, gz = (nx + 7) // 8, (ny + 7) // 8, (nz + 7) // 8
    st = dict(P=P, nVoxel=(nx, ny, nz), sVoxel=(sx, sy, sz), center=(cx, cy, cz), grid=(gx, gy, gz))
    st["vol"] = np.zeros((nx, ny, nz), _f)
    for k in ("radii_x", "radii_y", "radii_z"):
        st[k] = np.zeros(P, np.int32)
    R = L.r2ref_voxel_forward(C.c_int(P), C.c_int(nx), C.c_int(ny), C.c_int(nz), C.c_float(sx), C.c_float(sy),
                              C.c_float(sz), C.c_float(cx), C.c_float(cy), C.c_float(cz), _p(means3D), _p(opac),
                              _p(scales), C.c_float(scale_modifier), _p(rotations), _p(cov), _p(st["vol"]),
                              _p(st["radii_x"]), _p(st["radii_y"]), _p(st["radii_z"]))
    st["num_rendered"] = int(R)
    spec = [("depths", 0, (P,), _f), ("means3D_norm", 1, (P, 3), _f), ("cov3D", 2, (P, 6), _f),
            ("conic_opacity", 3, (P, 7), _f), ("tiles_touched", 5, (P,), np.uint32), ("offsets", 6, (P,), np.uint32),
            ("keys_unsorted", 7, (R,), np.uint64), ("vals_unsorted", 8, (R,), np.uint32), ("keys", 9, (R,), np.uint64),
            ("point_list", 10, (R,), np.uint32), ("ranges", 11, (gx * gy * gz, 2), np.uint32),
            ("n_contrib", 12, (nx * ny * nz,), np.uint32)]
    for name, which, shape, dt in spec:
        a = np.zeros(shape, dt)
        if P:
            L.r2ref_voxel_get(C.c_int(which), _p(a))
        st[name] = a
    return st


def voxel_backward(st, means3D, scales, rotations, scale_modifier, cov3D_precomp, dL_dvol):
    _, L = _libs()
    P = st["P"]
    sx, sy, sz = st["sVoxel"]
    cx, cy, cz = st["center"]
    means3D = np.ascontiguousarray(np.asarray(means3D, _f).reshape(-1, 3))
    scales, rotations, cov = _c32(scales), _c32(rotations), _c32(cov3D_precomp)
    dL = _c32(dL_dvol, (-1,))
    g = dict(dL_dmeans3D=np.zeros((P, 3), _f), dL_dmeans3D_norm=np.zeros((P, 3), _f), dL_dconic3D=np.zeros((P, 6), _f),
             dL_dopacity=np.zeros((P, 1), _f), dL_dcov3D=np.zeros((P, 6), _f), dL_dscales=np.zeros((P, 3), _f),
             dL_drotations=np.zeros((P, 4), _f))
    L.r2ref_voxel_backward(C.c_float(sx), C.c_float(sy), C.c_float(sz), C.c_float(cx), C.c_float(cy), C.c_float(cz),
                           _p(means3D), _p(scales), C.c_float(scale_modifier), _p(rotations), _p(cov), _p(st["radii_x"]),
                           _p(st["radii_y"]), _p(st["radii_z"]), _p(dL), _p(g["dL_dmeans3D_norm"]), _p(g["dL_dconic3D"]),
                           _p(g["dL_dopacity"]), _p(g["dL_dmeans3D"]), _p(g["dL_dcov3D"]), _p(g["dL_dscales"]),
                           _p(g["dL_drotations"]))
    return g
