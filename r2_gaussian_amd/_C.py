"""Python mirror of the reference's pybind module ``xray_gaussian_rasterization_voxelization._C``
(SUB/ext.cpp:17-23): the same five functions with the same argument lists and return tuples, implemented on
top of the C ABI of libr2hip.so.  This file plays the role of the torch boundary
SUB/rasterize_points.cu / SUB/voxelize_points.cu: it owns tensor allocation (outputs, zeroed gradient
buffers, the three opaque uint8 state tensors handed to the kernels through allocation callbacks) and passes
raw device pointers + the current HIP stream down.
"""
import os

import threading

import torch

from . import _lib

_F32 = torch.float32

# The same torch boundary compiled (csrc/torch_shim.cpp -> _r2shim.so, built by r2_gaussian_amd.build): ~70 us less
# interpreter time per training view than the ctypes path below.  Both drive the same libr2hip.so; R2_SHIM=0 forces ctypes.
_SHIM = None
_SHIM_TRIED = False
_POISON = os.environ.get("R2_POISON", "0") == "1"   # ctypes boundary only (R2_SHIM=0)


def _shim():
    """The compiled boundary, loaded on first use; None when it is not built, switched off, or an experiment library is
    selected with R2HIP_LIB (the module is linked against the product libr2hip.so)."""
    global _SHIM, _SHIM_TRIED
    if not _SHIM_TRIED:
        _SHIM_TRIED = True
        here = os.path.dirname(os.path.abspath(__file__))
        if (os.environ.get("R2_SHIM", "1") != "0" and not os.environ.get("R2HIP_LIB")
                and os.path.exists(os.path.join(here, "_r2shim.so"))):
            _lib.lib()   # load libr2hip.so first (and fail loudly if it is missing)
            from . import _r2shim
            _SHIM = _r2shim
    return _SHIM


def _raw_stream(dev):
    return torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device())


def _shim_call(fn, *args):
    try:
        return fn(*args)
    except RuntimeError as e:   # the library's error text, as the ctypes path raises it
        raise _lib.R2HipError(str(e)) from None


def _ptr(t):
    """Device pointer of a contiguous float32/int32 tensor; empty tensors become NULL (the reference passes
    ``torch.Tensor([])`` for absent inputs, whose data pointer is null)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def _dev_f32(t, like):
    """Contiguous float32 view on the kernel's device (SUB/rasterize_points.cu:79-93 ``.contiguous()``)."""
    if t is None or t.numel() == 0:
        return None
    if t.dtype != _F32:
        t = t.to(_F32)
    if t.device != like.device:
        t = t.to(like.device)
    t = t.contiguous()
    # the kernels read rotations / dL_dpix rows 16 bytes at a time (include/r2hip.h, "Alignment"): a view carved out of a
    # flat parameter buffer at an odd offset is contiguous but not 16-byte aligned -> take an (allocator-aligned) copy
    if t.data_ptr() & 15:
        t = t.clone()
    return t


def _require_gpu(t, name):
    if not t.is_cuda:
        raise _lib.R2HipError("%s must be a GPU tensor: the MI355X kernels have no CPU fallback" % name)


class _State:
    """The three opaque state buffers of one forward call, filled through the allocation callbacks (SUB/utility.h:7-13).

    The ctypes callback objects are expensive to create (tens of microseconds) and every forward call needs three, so
    they are created once per device and write into whichever ``_State`` is current FOR THE CALLING THREAD: a forward call
    is synchronous on its host thread (it returns num_rendered), but ctypes releases the GIL inside the call, so several
    threads (multistream.StreamPool) can be inside a forward on the same device at once -- the current state is
    thread-local (a callback runs on the thread that made the C call)."""

    __slots__ = ("bufs",)

    def __init__(self):
        self.bufs = [None, None, None]


class _DeviceHooks:
    def __init__(self, device):
        self.device = device
        self._tls = threading.local()
        self.empty = torch.empty(0, dtype=torch.uint8, device=device)
        self.cbs = [_lib.ALLOC_FN(self._make(i)) for i in range(3)]

    def _make(self, i):
        def alloc(nbytes, _user):
            try:
                # the state sizes follow num_rendered, which changes from view to view: round large requests up to a
                # coarse grid so that the caching allocator sees a handful of recurring sizes instead of a new one per
                # view (a miss is a hipMalloc, i.e. a device synchronisation in the middle of the forward pass)
                n = int(nbytes)
                if n > (1 << 20):
                    g = 1 << max(20, n.bit_length() - 4)   # 1/16 .. 1/8 of the size
                    n = (n + g - 1) // g * g
                if _POISON:   # debugging aid: 0xFF-filled state, so that a read of never-written state shows up
                    t = torch.full((n,), 255, dtype=torch.uint8, device=self.device)
                else:
                    t = torch.empty(n, dtype=torch.uint8, device=self.device)
                self._tls.current.bufs[i] = t
                return t.data_ptr()
            except Exception:   # out of memory etc.: report NULL, the C side turns it into R2_ERR_ALLOC
                return None
        return alloc

    def begin(self):
        st = _State()
        self._tls.current = st
        return st

    def finish(self, st):
        self._tls.current = None
        return [b if b is not None else self.empty for b in st.bufs]


_HOOKS = {}


def _hooks(device):
    h = _HOOKS.get(device)
    if h is None:
        h = _HOOKS[device] = _DeviceHooks(device)
    return h


class _on_device:
    """``with torch.cuda.device(dev)`` only when dev is not already current (the context manager costs ~10 us)."""

    __slots__ = ("ctx",)

    def __init__(self, dev):
        self.ctx = None if torch.cuda.current_device() == (dev.index or 0) else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def rasterize_gaussians(means3D, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                        projmatrix, tan_fovx, tan_fovy, image_height, image_width, campos, prefiltered, mode,
                        debug):
    """-> (num_rendered, out_color[1,H,W], radii[P] i32, geomBuffer u8, binningBuffer u8, imgBuffer u8)
    (SUB/rasterize_points.cu:28-97)."""
    if means3D.ndim != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    _require_gpu(means3D, "means3D")
    dev = means3D.device
    sh = _shim()
    if sh is not None:
        return _shim_call(sh.rasterize_gaussians, means3D, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                          viewmatrix, projmatrix, tan_fovx, tan_fovy, int(image_height), int(image_width), campos,
                          bool(prefiltered), int(mode), bool(debug), _raw_stream(dev))
    P, H, W = means3D.shape[0], int(image_height), int(image_width)
    hk = _hooks(dev)
    if P == 0:   # SUB/rasterize_points.cu:58-70: zero image, no state
        return 0, torch.zeros((1, H, W), dtype=_F32, device=dev), torch.zeros((0,), dtype=torch.int32, device=dev), \
            hk.empty, hk.empty, hk.empty
    # both outputs are written in full by the kernels (every pixel, every Gaussian): no zero-fill needed
    out_color = torch.empty((1, H, W), dtype=_F32, device=dev)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    m3 = _dev_f32(means3D, means3D)
    op, sc, ro, cp = (_dev_f32(t, means3D) for t in (opacity, scales, rotations, cov3D_precomp))
    vm, pm, cam = (_dev_f32(t, means3D) for t in (viewmatrix, projmatrix, campos))
    st = hk.begin()
    try:
        with _on_device(dev):
            rc = _lib.lib().r2_raster_forward(
                hk.cbs[0], None, hk.cbs[1], None, hk.cbs[2], None, P, W, H, _ptr(m3), _ptr(op), _ptr(sc),
                float(scale_modifier), _ptr(ro), _ptr(cp), _ptr(vm), _ptr(pm), _ptr(cam), float(tan_fovx),
                float(tan_fovy), int(bool(prefiltered)), int(mode), out_color.data_ptr(), radii.data_ptr(),
                int(bool(debug)), _stream(dev))
    finally:
        bufs = hk.finish(st)
    rendered = _lib.check(rc, "r2_raster_forward")
    return rendered, out_color, radii, bufs[0], bufs[1], bufs[2]


def rasterize_gaussians_backward(means3D, radii, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                                 projmatrix, tan_fovx, tan_fovy, dL_dout_color, campos, geomBuffer, R,
                                 binningBuffer, imageBuffer, mode, debug):
    """-> (dL_dmeans2D[P,3], dL_dopacity[P,1], dL_dmu[P,1], dL_dmeans3D[P,3], dL_dcov3D[P,6], dL_dscales[P,3],
    dL_drotations[P,4])  (SUB/rasterize_points.cu:99-164)."""
    _require_gpu(means3D, "means3D")
    dev = means3D.device
    sh = _shim()
    if sh is not None:
        return _shim_call(sh.rasterize_gaussians_backward, means3D, radii, scales, rotations, scale_modifier,
                          cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, campos, geomBuffer,
                          int(R), binningBuffer, imageBuffer, int(mode), bool(debug), _raw_stream(dev))
    P = means3D.shape[0]
    H, W = int(dL_dout_color.shape[1]), int(dL_dout_color.shape[2])
    # one allocation for all eight gradient arrays (25 floats per Gaussian), 16-byte rows first; the kernels write
    # every row (zeros for culled Gaussians), so no fill
    flat = torch.empty(25 * P, dtype=_F32, device=dev)
    # 8 arrays carved out of `flat` (16-byte rows first) with one as_strided each -- this function runs on the autograd
    # engine's critical path, every avoided tensor op is ~2 us of host time per training view
    # order: conic and rot first (their rows are written 16 bytes at a time), then the four parameter gradients a trainer
    # exchanges between GPUs ADJACENT to each other -- rot | means3D | scales | opacity = one contiguous [11 P] block that
    # dist.grad_block() hands to the all-reduce without a packing copy
    o = 0
    dL_dconic = flat.as_strided((P, 2, 2), (4, 2, 1), o); o += 4 * P
    dL_drot = flat.as_strided((P, 4), (4, 1), o); o += 4 * P
    dL_dmeans3D = flat.as_strided((P, 3), (3, 1), o); o += 3 * P
    dL_dscales = flat.as_strided((P, 3), (3, 1), o); o += 3 * P
    dL_dopacity = flat.as_strided((P, 1), (1, 1), o); o += P
    dL_dmeans2D = flat.as_strided((P, 3), (3, 1), o); o += 3 * P
    dL_dmu = flat.as_strided((P, 1), (1, 1), o); o += P
    dL_dcov3D = flat.as_strided((P, 6), (6, 1), o)
    if P != 0:
        m3 = _dev_f32(means3D, means3D)
        sc, ro, cp = (_dev_f32(t, means3D) for t in (scales, rotations, cov3D_precomp))
        vm, pm, cam = (_dev_f32(t, means3D) for t in (viewmatrix, projmatrix, campos))
        g = _dev_f32(dL_dout_color, means3D)
        rad = radii.contiguous()
        with _on_device(dev):
            rc = _lib.lib().r2_raster_backward(
                P, int(R), W, H, _ptr(m3), _ptr(sc), float(scale_modifier), _ptr(ro), _ptr(cp), _ptr(vm), _ptr(pm),
                _ptr(cam), float(tan_fovx), float(tan_fovy), rad.data_ptr(), _ptr(geomBuffer), _ptr(binningBuffer),
                _ptr(imageBuffer), _ptr(g), dL_dmeans2D.data_ptr(), dL_dconic.data_ptr(), dL_dopacity.data_ptr(),
                dL_dmu.data_ptr(), dL_dmeans3D.data_ptr(), dL_dcov3D.data_ptr(), dL_dscales.data_ptr(),
                dL_drot.data_ptr(), int(mode), int(bool(debug)), _stream(dev))
        _lib.check(rc, "r2_raster_backward")
    return dL_dmeans2D, dL_dopacity, dL_dmu, dL_dmeans3D, dL_dcov3D, dL_dscales, dL_drot


# ---------------------------------------------------------------------------------------------- batched views (new)
def rasterize_gaussians_batch(means3D, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrices, projmatrices,
                              tan_fovx, tan_fovy, image_height, image_width, mode, debug):
    """V views of the same Gaussians in one pass (r2_raster_forward_batch): viewmatrices / projmatrices [V,4,4] ->
    (num_rendered, out_color[V,H,W], radii[V,P] i32, geomBuffer, binningBuffer, imgBuffer).  Each view's image and radii are
    bit-identical to ``rasterize_gaussians`` for that view."""
    if means3D.ndim != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    _require_gpu(means3D, "means3D")
    if viewmatrices.ndim != 3 or viewmatrices.shape[1:] != (4, 4) or projmatrices.shape != viewmatrices.shape:
        raise RuntimeError("viewmatrices / projmatrices must have dimensions (num_views, 4, 4)")
    dev = means3D.device
    P, V, H, W = means3D.shape[0], viewmatrices.shape[0], int(image_height), int(image_width)
    hk = _hooks(dev)
    if P == 0:
        return 0, torch.zeros((V, H, W), dtype=_F32, device=dev), torch.zeros((V, 0), dtype=torch.int32, device=dev), \
            hk.empty, hk.empty, hk.empty
    out_color = torch.empty((V, H, W), dtype=_F32, device=dev)
    radii = torch.empty((V, P), dtype=torch.int32, device=dev)
    m3 = _dev_f32(means3D, means3D)
    op, sc, ro, cp = (_dev_f32(t, means3D) for t in (opacity, scales, rotations, cov3D_precomp))
    vm, pm = (_dev_f32(t, means3D) for t in (viewmatrices, projmatrices))
    st = hk.begin()
    try:
        with _on_device(dev):
            rc = _lib.lib().r2_raster_forward_batch(
                hk.cbs[0], None, hk.cbs[1], None, hk.cbs[2], None, P, V, W, H, _ptr(m3), _ptr(op), _ptr(sc),
                float(scale_modifier), _ptr(ro), _ptr(cp), _ptr(vm), _ptr(pm), float(tan_fovx), float(tan_fovy), int(mode),
                out_color.data_ptr(), radii.data_ptr(), int(bool(debug)), _stream(dev))
    finally:
        bufs = hk.finish(st)
    rendered = _lib.check(rc, "r2_raster_forward_batch")
    return rendered, out_color, radii, bufs[0], bufs[1], bufs[2]


def rasterize_gaussians_backward_batch(means3D, radii, scales, rotations, scale_modifier, cov3D_precomp, viewmatrices,
                                       projmatrices, tan_fovx, tan_fovy, dL_dout_color, geomBuffer, R, binningBuffer,
                                       imageBuffer, mode, debug):
    """-> (dL_dmeans2D[V,P,3], dL_dopacity[P,1], dL_dmu[V,P], dL_dmeans3D[P,3], dL_dcov3D[P,6], dL_dscales[P,3],
    dL_drotations[P,4]): per-view screen-space gradients (the densification statistics are per view), parameter gradients
    summed over the V views in view order."""
    _require_gpu(means3D, "means3D")
    dev = means3D.device
    P, V = means3D.shape[0], viewmatrices.shape[0]
    H, W = int(dL_dout_color.shape[-2]), int(dL_dout_color.shape[-1])
    # per-view arrays first (16-byte rows in front), then rot | means3D | scales | opacity adjacent like the single-view
    # backward: the block dist.grad_block() hands to the all-reduce
    flat = torch.empty(8 * V * P + 17 * P, dtype=_F32, device=dev)
    o = 0
    dL_dconic = flat.as_strided((V, P, 2, 2), (4 * P, 4, 2, 1), o); o += 4 * V * P
    dL_drot = flat.as_strided((P, 4), (4, 1), o); o += 4 * P
    dL_dmeans3D = flat.as_strided((P, 3), (3, 1), o); o += 3 * P
    dL_dscales = flat.as_strided((P, 3), (3, 1), o); o += 3 * P
    dL_dopacity = flat.as_strided((P, 1), (1, 1), o); o += P
    dL_dcov3D = flat.as_strided((P, 6), (6, 1), o); o += 6 * P
    dL_dmeans2D = flat.as_strided((V, P, 3), (3 * P, 3, 1), o); o += 3 * V * P
    dL_dmu = flat.as_strided((V, P), (P, 1), o)
    if P != 0:
        m3 = _dev_f32(means3D, means3D)
        sc, ro, cp = (_dev_f32(t, means3D) for t in (scales, rotations, cov3D_precomp))
        vm, pm = (_dev_f32(t, means3D) for t in (viewmatrices, projmatrices))
        g = _dev_f32(dL_dout_color, means3D)
        rad = radii.contiguous()
        with _on_device(dev):
            rc = _lib.lib().r2_raster_backward_batch(
                P, V, int(R), W, H, _ptr(m3), _ptr(sc), float(scale_modifier), _ptr(ro), _ptr(cp), _ptr(vm), _ptr(pm),
                float(tan_fovx), float(tan_fovy), rad.data_ptr(), _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer),
                _ptr(g), dL_dmeans2D.data_ptr(), dL_dconic.data_ptr(), dL_dopacity.data_ptr(), dL_dmu.data_ptr(),
                dL_dmeans3D.data_ptr(), dL_dcov3D.data_ptr(), dL_dscales.data_ptr(), dL_drot.data_ptr(), int(mode),
                int(bool(debug)), _stream(dev))
        _lib.check(rc, "r2_raster_backward_batch")
    return dL_dmeans2D, dL_dopacity, dL_dmu, dL_dmeans3D, dL_dcov3D, dL_dscales, dL_drot


def mark_visible(means3D, viewmatrix, projmatrix):
    """-> bool[P], ``z_view > 0.2``  (SUB/rasterize_points.cu:166-185)."""
    _require_gpu(means3D, "means3D")
    dev = means3D.device
    P = means3D.shape[0]
    present = torch.zeros((P,), dtype=torch.bool, device=dev)
    if P != 0:
        m3, vm, pm = (_dev_f32(t, means3D) for t in (means3D, viewmatrix, projmatrix))
        with _on_device(dev):
            rc = _lib.lib().r2_mark_visible(P, _ptr(m3), _ptr(vm), _ptr(pm), present.data_ptr(), _stream(dev))
        _lib.check(rc, "r2_mark_visible")
    return present


TILE3D = 8


def _slab_tiles(nVoxel_x, tile_x0, tile_x1):
    """(tile_x0, tile_x1, voxels of the slab along x); tile_x1 None / negative = up to the last layer."""
    layers = (int(nVoxel_x) + TILE3D - 1) // TILE3D
    t0 = int(tile_x0)
    t1 = layers if tile_x1 is None or int(tile_x1) < 0 else int(tile_x1)
    if not (0 <= t0 < t1 <= layers):
        raise _lib.R2HipError("x-slab [%d, %d) is not a non-empty range of the grid's %d tile layers" % (t0, t1, layers))
    return t0, t1, min(t1 * TILE3D, int(nVoxel_x)) - t0 * TILE3D


def voxelize_gaussians(means3D, opacity, scales, rotations, scale_modifier, cov3D_precomp, nVoxel_x, nVoxel_y,
                       nVoxel_z, sVoxel_x, sVoxel_y, sVoxel_z, center_x, center_y, center_z, prefiltered, debug):
    """-> (num_rendered, out_volume[nx,ny,nz], radii_x, radii_y, radii_z, geomBuffer, binningBuffer, imgBuffer)
    (SUB/voxelize_points.cu:29-98)."""
    return voxelize_gaussians_slab(means3D, opacity, scales, rotations, scale_modifier, cov3D_precomp, nVoxel_x, nVoxel_y,
                                   nVoxel_z, sVoxel_x, sVoxel_y, sVoxel_z, center_x, center_y, center_z, prefiltered, debug,
                                   0, None)


def voxelize_gaussians_slab(means3D, opacity, scales, rotations, scale_modifier, cov3D_precomp, nVoxel_x, nVoxel_y,
                            nVoxel_z, sVoxel_x, sVoxel_y, sVoxel_z, center_x, center_y, center_z, prefiltered, debug,
                            tile_x0, tile_x1):
    """``voxelize_gaussians`` for the tile layers [tile_x0, tile_x1) along x of the grid the other arguments describe
    (r2_voxel_forward_slab; new: the unit of the sharded query): out_volume is the slab's [x1 - x0, ny, nz] block, bit-identical
    to those voxels of the full call."""
    if means3D.ndim != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    _require_gpu(means3D, "means3D")
    dev = means3D.device
    t0, t1, nxs = _slab_tiles(nVoxel_x, tile_x0, tile_x1)
    sh = _shim()
    if sh is not None:
        return _shim_call(sh.voxelize_gaussians, means3D, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                          int(nVoxel_x), int(nVoxel_y), int(nVoxel_z), float(sVoxel_x), float(sVoxel_y), float(sVoxel_z),
                          float(center_x), float(center_y), float(center_z), bool(prefiltered), bool(debug), _raw_stream(dev),
                          t0, t1)
    P = means3D.shape[0]
    nx, ny, nz = int(nVoxel_x), int(nVoxel_y), int(nVoxel_z)
    hk = _hooks(dev)
    if P == 0:
        z = torch.zeros((0,), dtype=torch.int32, device=dev)
        return 0, torch.zeros((nxs, ny, nz), dtype=_F32, device=dev), z, z.clone(), z.clone(), hk.empty, hk.empty, hk.empty
    out = torch.empty((nxs, ny, nz), dtype=_F32, device=dev)      # written in full by the combine kernel
    radii = torch.empty((3, P), dtype=torch.int32, device=dev)    # written in full by the preprocess kernel
    m3 = _dev_f32(means3D, means3D)
    op, sc, ro, cp = (_dev_f32(t, means3D) for t in (opacity, scales, rotations, cov3D_precomp))
    st = hk.begin()
    try:
        with _on_device(dev):
            rc = _lib.lib().r2_voxel_forward_slab(
                hk.cbs[0], None, hk.cbs[1], None, hk.cbs[2], None, P, nx, ny, nz, float(sVoxel_x), float(sVoxel_y),
                float(sVoxel_z), float(center_x), float(center_y), float(center_z), t0, t1, _ptr(m3), _ptr(op), _ptr(sc),
                float(scale_modifier), _ptr(ro), _ptr(cp), int(bool(prefiltered)), out.data_ptr(),
                radii[0].data_ptr(), radii[1].data_ptr(), radii[2].data_ptr(), int(bool(debug)), _stream(dev))
    finally:
        bufs = hk.finish(st)
    rendered = _lib.check(rc, "r2_voxel_forward_slab")
    return rendered, out, radii[0], radii[1], radii[2], bufs[0], bufs[1], bufs[2]


def voxelize_gaussians_backward(means3D, radii_x, radii_y, radii_z, scales, rotations, scale_modifier,
                                cov3D_precomp, dL_dout_color, geomBuffer, R, binningBuffer, imageBuffer, nVoxel_x,
                                nVoxel_y, nVoxel_z, sVoxel_x, sVoxel_y, sVoxel_z, center_x, center_y, center_z,
                                debug):
    """-> (dL_dopacity[P,1], dL_dmeans3D[P,3], dL_dcov3D[P,6], dL_dscales[P,3], dL_drotations[P,4])
    (SUB/voxelize_points.cu:102-167)."""
    return voxelize_gaussians_backward_slab(means3D, radii_x, radii_y, radii_z, scales, rotations, scale_modifier,
                                            cov3D_precomp, dL_dout_color, geomBuffer, R, binningBuffer, imageBuffer, nVoxel_x,
                                            nVoxel_y, nVoxel_z, sVoxel_x, sVoxel_y, sVoxel_z, center_x, center_y, center_z,
                                            debug, 0, None)


def voxelize_gaussians_backward_slab(means3D, radii_x, radii_y, radii_z, scales, rotations, scale_modifier,
                                     cov3D_precomp, dL_dout_color, geomBuffer, R, binningBuffer, imageBuffer, nVoxel_x,
                                     nVoxel_y, nVoxel_z, sVoxel_x, sVoxel_y, sVoxel_z, center_x, center_y, center_z,
                                     debug, tile_x0, tile_x1):
    """Backward of ``voxelize_gaussians_slab`` (dL_dout_color = the slab's block)."""
    _require_gpu(means3D, "means3D")
    dev = means3D.device
    t0, t1, _nxs = _slab_tiles(nVoxel_x, tile_x0, tile_x1)
    sh = _shim()
    if sh is not None:
        return _shim_call(sh.voxelize_gaussians_backward, means3D, radii_x, radii_y, radii_z, scales, rotations,
                          scale_modifier, cov3D_precomp, dL_dout_color, geomBuffer, int(R), binningBuffer, imageBuffer,
                          int(nVoxel_x), int(nVoxel_y), int(nVoxel_z), float(sVoxel_x), float(sVoxel_y), float(sVoxel_z),
                          float(center_x), float(center_y), float(center_z), bool(debug), _raw_stream(dev), t0, t1)
    P = means3D.shape[0]
    flat = torch.empty(26 * P, dtype=_F32, device=dev)   # every row is written by the kernels (zeros where culled)
    cuts = [4, 3, 3, 6, 1, 6, 3]
    views, o = [], 0
    for c in cuts:
        views.append(flat[o:o + c * P])
        o += c * P
    dL_drot = views[0].view(P, 4)
    dL_dmeans3D = views[1].view(P, 3)
    dL_dmeans3D_norm = views[2].view(P, 3)
    dL_dconic3D = views[3].view(P, 6)
    dL_dopacity = views[4].view(P, 1)
    dL_dcov3D = views[5].view(P, 6)
    dL_dscales = views[6].view(P, 3)
    if P != 0:
        m3 = _dev_f32(means3D, means3D)
        sc, ro, cp = (_dev_f32(t, means3D) for t in (scales, rotations, cov3D_precomp))
        g = _dev_f32(dL_dout_color, means3D)
        rx, ry, rz = radii_x.contiguous(), radii_y.contiguous(), radii_z.contiguous()
        with _on_device(dev):
            rc = _lib.lib().r2_voxel_backward_slab(
                P, int(R), int(nVoxel_x), int(nVoxel_y), int(nVoxel_z), float(sVoxel_x), float(sVoxel_y),
                float(sVoxel_z), float(center_x), float(center_y), float(center_z), t0, t1, _ptr(m3), _ptr(sc),
                float(scale_modifier), _ptr(ro), _ptr(cp), rx.data_ptr(), ry.data_ptr(), rz.data_ptr(),
                _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer), _ptr(g), dL_dmeans3D_norm.data_ptr(),
                dL_dconic3D.data_ptr(), dL_dopacity.data_ptr(), dL_dmeans3D.data_ptr(), dL_dcov3D.data_ptr(),
                dL_dscales.data_ptr(), dL_drot.data_ptr(), int(bool(debug)), _stream(dev))
        _lib.check(rc, "r2_voxel_backward_slab")
    return dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dscales, dL_drot


def distCUDA2(points):
    """simple_knn._C.distCUDA2: mean squared distance to the 3 nearest neighbours, [P] f32."""
    _require_gpu(points, "points")
    dev = points.device
    pts = _dev_f32(points, points)
    P = points.shape[0]
    out = torch.zeros((P,), dtype=_F32, device=dev)
    if P != 0:
        L = _lib.lib()
        # the library does not allocate: the grid search's workspace comes from torch's allocator (r2hip.h)
        nbytes = int(L.r2_knn_workspace_bytes(P))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        with _on_device(dev):
            rc = L.r2_knn_dist2_ws(P, _ptr(pts), out.data_ptr(), ws.data_ptr(), nbytes, _stream(dev))
        _lib.check(rc, "r2_knn_dist2_ws")
    return out
