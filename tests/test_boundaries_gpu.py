"""Both torch boundaries of the C ABI drive the same kernels: the compiled one (csrc/torch_shim.cpp -> _r2shim.so, the default)
and the ctypes one (r2_gaussian_amd/_C.py, R2_SHIM=0).  Each is run through forward + backward of the rasterizer and the
voxelizer against the oracle, its gradient carve is checked to be the contiguous [11 P] block dist.grad_block() relies on, and
mis-aligned inputs (views into a flat parameter buffer at an odd offset) are handled."""
import numpy as np
import pytest
import torch

from r2_gaussian_amd import scene as S
from tests import helpers as Hh

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["shim", "ctypes"])
def boundary(request, monkeypatch):
    from r2_gaussian_amd import _C
    if request.param == "ctypes":
        monkeypatch.setattr(_C, "_SHIM", None)
        monkeypatch.setattr(_C, "_SHIM_TRIED", True)
        assert _C._shim() is None
    else:
        monkeypatch.setattr(_C, "_SHIM_TRIED", False)
        monkeypatch.setenv("R2_SHIM", "1")
        monkeypatch.delenv("R2HIP_LIB", raising=False)
        if _C._shim() is None:
            pytest.skip("_r2shim.so not built")
    return request.param


def test_raster_and_voxel_through_either_boundary(boundary, oracle, gpu):
    from r2_gaussian_amd import dist as D
    c = S.make_cloud(6000, seed=12)
    v = S.make_view(0.9, (80, 96))
    o = Hh.oracle_raster(oracle, c, v)
    h = Hh.hip_raster(c, v, gpu)
    assert h["num_rendered"] == o["num_rendered"] and np.array_equal(h["radii"], o["radii"])
    Hh.check_binning(h, o)
    Hh.parity_image(oracle, o, h["color"], "boundary=%s raster" % boundary)
    dL = S.make_pixel_grad(80, 96).numpy()
    from r2_gaussian_amd import _C
    a = h["args"]
    geom, binning, img = h["bufs"]
    res = _C.rasterize_gaussians_backward(a[0], torch.as_tensor(h["radii"]).to(gpu), a[2], a[3], a[4], a[5], a[6], a[7], a[8],
                                          a[9], torch.as_tensor(dL).to(gpu), a[12], geom, h["num_rendered"], binning, img,
                                          v.mode, False)
    torch.cuda.synchronize()
    names = ["dL_dmeans2D", "dL_dopacity", "dL_dmu", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations"]
    gh = {n: t.cpu().numpy() for n, t in zip(names, res)}
    Hh.parity_raster_grads(oracle, o, gh, c, v, dL, "boundary=%s raster" % boundary)
    # the four parameter gradients ARE one contiguous [11 P] block (rotation | xyz | scaling | density): in-place all-reduce
    g = dict(zip(names, res))
    blk = D.grad_block(g["dL_dmeans3D"], g["dL_dopacity"], g["dL_dscales"], g["dL_drotations"])
    assert blk is not None and blk.numel() == 11 * 6000 and blk.data_ptr() == g["dL_drotations"].data_ptr()
    # voxelizer
    n, s, ctr = (32, 40, 24), (1.0, 1.25, 0.75), (0.05, -0.1, 0.0)
    ov = Hh.oracle_voxel(oracle, c, n, s, ctr)
    hv = Hh.hip_voxel(c, n, s, ctr, gpu)
    Hh.check_binning(hv, ov)
    Hh.parity_volume(oracle, ov, hv["vol"], "boundary=%s voxel" % boundary)
    gen = torch.Generator().manual_seed(3)
    dLv = ((torch.rand(*n, generator=gen) * 2 - 1) / float(np.prod(n))).numpy()
    Hh.parity_voxel_grads(oracle, ov, Hh.hip_voxel_backward(hv, c, n, s, ctr, dLv, gpu), c, dLv, "boundary=%s voxel" % boundary)


def test_misaligned_inputs_are_copied(boundary, oracle, gpu):
    """rotations carved out of a flat buffer at an offset that is not a multiple of 16 bytes (P % 4 != 0 in front of it)."""
    from r2_gaussian_amd import _C
    P = 1002   # (3 P + 1) * 4 bytes is not a multiple of 16
    c = S.make_cloud(P, seed=2)
    v = S.make_view(0.2, (64, 64))
    flat = torch.empty(3 * P + 1 + 4 * P + 3 * P + P, device=gpu)
    xyz = flat[1:1 + 3 * P].view(P, 3)                       # 4 bytes off
    rot = flat[3 * P + 1:3 * P + 1 + 4 * P].view(P, 4)       # (3 P + 1) floats in: not 16-byte aligned
    scal = flat[7 * P + 1:10 * P + 1].view(P, 3)
    dens = flat[10 * P + 1:11 * P + 1].view(P, 1)
    xyz.copy_(c.xyz); rot.copy_(c.rotations); scal.copy_(c.scales); dens.copy_(c.density)
    assert rot.data_ptr() % 16 != 0 and rot.is_contiguous()
    e = torch.empty(0)
    R, color, radii, g, b, i = _C.rasterize_gaussians(xyz, dens, scal, rot, 1.0, e, v.world_view_transform.to(gpu),
                                                      v.full_proj_transform.to(gpu), v.tanfovx, v.tanfovy, 64, 64,
                                                      v.camera_center.to(gpu), False, v.mode, False)
    o = Hh.oracle_raster(oracle, c, v)
    assert R == o["num_rendered"] and np.array_equal(radii.cpu().numpy(), o["radii"])
    Hh.parity_image(oracle, o, color.cpu().numpy(), "boundary=%s misaligned" % boundary)
    dL = S.make_pixel_grad(64, 64).to(gpu)
    res = _C.rasterize_gaussians_backward(xyz, radii, scal, rot, 1.0, e, v.world_view_transform.to(gpu),
                                          v.full_proj_transform.to(gpu), v.tanfovx, v.tanfovy, dL, v.camera_center.to(gpu),
                                          g, R, b, i, v.mode, False)
    torch.cuda.synchronize()
    names = ["dL_dmeans2D", "dL_dopacity", "dL_dmu", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations"]
    Hh.parity_raster_grads(oracle, o, {n: t.cpu().numpy() for n, t in zip(names, res)}, c, v, dL.cpu().numpy(),
                           "boundary=%s misaligned" % boundary)


@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_one_thread_two_devices_unhinted(oracle, gpu):
    """One host thread drives cuda:0 then cuda:1 with depth hints off (the read-back path with a per-device event)."""
    from r2_gaussian_amd import _lib
    L = _lib.lib()
    L.r2_depth_hint_control(0)
    try:
        c = S.make_cloud(3000, seed=8)
        v = S.make_view(1.0, (64, 64))
        o = Hh.oracle_raster(oracle, c, v)
        for d in (torch.device("cuda:0"), torch.device("cuda:1"), torch.device("cuda:0")):
            h = Hh.hip_raster(c, v, d)
            assert h["num_rendered"] == o["num_rendered"] and np.array_equal(h["point_list"], o["point_list"])
    finally:
        L.r2_depth_hint_control(1)
