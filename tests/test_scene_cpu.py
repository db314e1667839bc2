"""CPU host logic: the synthetic-scene module reproduces the reference's camera conventions (golden vectors made by
importing the reference's own Python, tests/golden/make_golden.py), the drop-in Python surface has the reference's
names / fields / error behaviour, and the 3D-PSNR definition matches ``metric_vol``."""
import os

import numpy as np
import pytest
import torch

from r2_gaussian_amd import scene as S

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("mode_name,scanner", [("cone", S.CONE_BEAM), ("parallel", S.PARALLEL_BEAM)])
def test_camera_matrices_match_reference(mode_name, scanner):
    z = np.load(os.path.join(GOLD, "camera.npz"))
    for i, a in enumerate(z["angles"]):
        v = S.make_view(float(a), (64, 64), scanner)
        assert np.array_equal(v.world_view_transform.numpy(), z[mode_name + "_world_view"][i])
        assert np.array_equal(v.full_proj_transform.numpy(), z[mode_name + "_full_proj"][i])
        np.testing.assert_allclose(v.camera_center.numpy(), z[mode_name + "_center"][i], rtol=0, atol=1e-6)
        np.testing.assert_allclose([v.tanfovx, v.tanfovy], z[mode_name + "_tanfov"][i], rtol=1e-12)


def test_view_angles_and_cloud_are_seeded():
    vs = S.make_views(50, (32, 32))
    assert len(vs) == 50 and vs[0].angle == 0.0 and abs(vs[25].angle - np.pi) < 1e-12
    a, b = S.make_cloud(1000, seed=0), S.make_cloud(1000, seed=0)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert a.xyz.shape == (1000, 3) and a.density.shape == (1000, 1) and a.rotations.shape == (1000, 4)
    np.testing.assert_allclose(a.rotations.norm(dim=1).numpy(), 1.0, rtol=1e-6)
    assert float(a.scales.min()) >= 0.001 and float(a.scales.max()) <= 1.0
    assert float(a.density.min()) >= 0.01 and float(a.density.max()) <= 0.3


def test_psnr3d_matches_reference_metric_vol():
    z = np.load(os.path.join(GOLD, "psnr.npz"))
    got = S.psnr3d(torch.from_numpy(z["vol_gt"]), torch.from_numpy(z["vol_pred"]))
    assert abs(got - float(z["psnr"])) < 1e-4


def test_dropin_surface_names_and_fields():
    """Same exports / NamedTuple field order as PY/__init__.py:1-2, PY/rasterization.py:200-211, PY/voxelization.py:26-38."""
    import xray_gaussian_rasterization_voxelization as X
    from simple_knn._C import distCUDA2   # noqa: F401
    assert X.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "scale_modifier", "viewmatrix", "projmatrix", "campos",
        "prefiltered", "mode", "debug")
    assert X.GaussianVoxelizationSettings._fields == (
        "scale_modifier", "nVoxel_x", "nVoxel_y", "nVoxel_z", "sVoxel_x", "sVoxel_y", "sVoxel_z", "center_x", "center_y",
        "center_z", "prefiltered", "debug")
    for f in ("rasterize_gaussians", "rasterize_gaussians_backward", "voxelize_gaussians", "voxelize_gaussians_backward",
              "mark_visible"):
        assert callable(getattr(X._C, f))


def test_exactly_one_of_scales_or_cov():
    """PY/rasterization.py:238-247 / PY/voxelization.py:240-249: exactly one of (scales, rotations) / cov3D_precomp."""
    from r2_gaussian_amd import (GaussianRasterizationSettings, GaussianRasterizer, GaussianVoxelizationSettings,
                                 GaussianVoxelizer)
    rs = GaussianRasterizationSettings(16, 16, 1.0, 1.0, 1.0, torch.eye(4), torch.eye(4), torch.zeros(3), False, 1, False)
    r = GaussianRasterizer(rs)
    x = torch.zeros(2, 3)
    with pytest.raises(Exception, match="exactly one"):
        r(x, x, torch.zeros(2, 1))
    with pytest.raises(Exception, match="exactly one"):
        r(x, x, torch.zeros(2, 1), scales=torch.ones(2, 3), rotations=torch.ones(2, 4), cov3D_precomp=torch.ones(2, 6))
    vs = GaussianVoxelizationSettings(1.0, 8, 8, 8, 2.0, 2.0, 2.0, 0.0, 0.0, 0.0, False, False)
    vz = GaussianVoxelizer(vs)
    with pytest.raises(Exception, match="exactly one"):
        vz(x, torch.zeros(2, 1))
