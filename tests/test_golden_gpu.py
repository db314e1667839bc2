"""GPU parity against the committed golden fixtures (tests/golden/*.npz = outputs of the reference's own kernels run
on the CPU, see tests/golden/make_golden.py): the HIP path, called through the C ABI, must reproduce the reference's
radii / sorted point lists / ranges bit for bit, images and volumes within 1e-4 relative, gradients within 2e-3 of each
array's scale (the reference's own gradient sums are float-atomic, order-nondeterministic)."""
import glob
import os

import numpy as np
import pytest
import torch

from r2_gaussian_amd import scene as S
from tests import helpers as Hh

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RASTER = sorted(glob.glob(os.path.join(GOLD, "raster_*.npz")))
VOXEL = sorted(glob.glob(os.path.join(GOLD, "voxel_*.npz")))


def _cloud(z):
    return S.Cloud(*(torch.from_numpy(z[k].copy()) for k in ("in_means3D", "in_scales", "in_rotations", "in_opacities")))


def _grad_close(name, got, want, frac=2e-3):
    scale = max(float(np.abs(want).max()), 1e-30)
    err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
    assert err <= frac * scale, "%s: max err %.3e vs scale %.3e" % (name, err, scale)


@pytest.mark.parametrize("path", RASTER, ids=[os.path.basename(p)[:-4] for p in RASTER])
def test_raster_vs_reference_golden(path, gpu):
    z = np.load(path)
    H, W, mode, precomp = (int(v) for v in z["in_meta"])
    tfx, tfy, mod = (float(v) for v in z["in_params"])
    c = _cloud(z)
    v = S.View(0.0, mode, H, W, tfx, tfy, torch.from_numpy(z["in_viewmatrix"].copy()),
               torch.from_numpy(z["in_projmatrix"].copy()), torch.zeros(3))
    cov = z["in_cov3D_precomp"] if precomp else None
    h = Hh.hip_raster(c, v, gpu, debug=True, cov3D_precomp=cov, scale_modifier=mod)
    assert h["num_rendered"] == int(z["num_rendered"])
    assert np.array_equal(h["radii"], z["fw_radii"])
    assert np.array_equal(h["tiles_touched"], z["fw_tiles_touched"])
    assert np.array_equal(h["keys"], z["fw_keys"]), "sorted (tile|depth) keys differ from the reference"
    assert np.array_equal(h["point_list"], z["fw_point_list"])
    assert np.array_equal(h["ranges"], z["fw_ranges"])
    ref = z["fw_color"]
    assert (np.abs(h["color"] - ref) <= 1e-4 * np.abs(ref) + 2e-5).all()
    # n_contrib: identical except where a pair sits on the alpha cut-off (exp2 vs exp rounding)
    assert (h["n_contrib"] != z["fw_n_contrib"]).mean() < 0.01
    g = Hh.hip_raster_backward(h, c, v, z["in_dL_dcolor"], gpu)
    for k in ("dL_dmeans2D", "dL_dopacity", "dL_dmu", "dL_dmeans3D", "dL_dcov3D"):
        _grad_close(k, g[k], z["bw_" + k])
    if not precomp:
        for k in ("dL_dscales", "dL_drotations"):
            _grad_close(k, g[k], z["bw_" + k])


@pytest.mark.parametrize("path", VOXEL, ids=[os.path.basename(p)[:-4] for p in VOXEL])
def test_voxel_vs_reference_golden(path, gpu):
    z = np.load(path)
    nV = tuple(int(v) for v in z["in_nVoxel"])
    sV, ctr = tuple(float(v) for v in z["in_sVoxel"]), tuple(float(v) for v in z["in_center"])
    mod = float(z["in_params"][0])
    c = _cloud(z)
    h = Hh.hip_voxel(c, nV, sV, ctr, gpu, debug=True, scale_modifier=mod)
    assert h["num_rendered"] == int(z["num_rendered"])
    for k in ("radii_x", "radii_y", "radii_z", "tiles_touched", "keys", "point_list", "ranges"):
        assert np.array_equal(h[k], z["fw_" + k]), k
    ref = z["fw_vol"]
    assert (np.abs(h["vol"] - ref) <= 1e-4 * np.abs(ref) + 2e-6).all()
    g = Hh.hip_voxel_backward(h, c, nV, sV, ctr, z["in_dL_dvol"], gpu)
    for k in ("dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations"):
        _grad_close(k, g[k], z["bw_" + k])
