"""The render kernels' hand-written EXEC-mask inline asm (v_cmpx + masked accumulation, csrc/raster_render.hip block_moments_lds,
csrc/voxel_render.hip vfwd_item) against the same kernels built with the plain C++ statement of that arithmetic
(-DR2_EXP_NO_CMPX -> r2_gaussian_amd/libr2hip_nocmpx.so, built by __graft_entry__.build()): identical results.  The variant
library is loaded in a subprocess (R2HIP_LIB), the product in this one."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT = os.path.join(ROOT, "r2_gaussian_amd", "libr2hip_nocmpx.so")

_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from r2_gaussian_amd import scene as S
from tests import helpers as Hh
dev = torch.device("cuda:0")
c = S.make_cloud(30000, seed=21)
v = S.make_views(8, (160, 144))[3]
h = Hh.hip_raster(c, v, dev)
dL = S.make_pixel_grad(160, 144).numpy()
g = Hh.hip_raster_backward(h, c, v, dL, dev)
n, s, ctr = (48, 40, 56), (1.5, 1.25, 1.75), (0.0, 0.05, -0.05)
hv = Hh.hip_voxel(c, n, s, ctr, dev)
np.savez(sys.argv[1], color=h["color"], vol=hv["vol"], **g)
"""


def _run(tmp_path, name, lib):
    out = str(tmp_path / (name + ".npz"))
    env = dict(os.environ)
    env.pop("R2HIP_LIB", None)
    if lib:
        env["R2HIP_LIB"] = lib
    r = subprocess.run([sys.executable, "-c", _SCRIPT % ROOT, out], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out)


def test_exec_mask_asm_equals_the_plain_cxx_build(tmp_path):
    if not os.path.exists(VARIANT):
        pytest.skip("libr2hip_nocmpx.so not built (run __graft_entry__.build())")
    a = _run(tmp_path, "product", None)
    b = _run(tmp_path, "nocmpx", VARIANT)
    for k in a.files:
        assert a[k].shape == b[k].shape, k
        scale = np.abs(a[k]).max()
        assert np.abs(a[k].astype(np.float64) - b[k]).max() <= 1e-6 * scale, (k, np.abs(a[k] - b[k]).max(), scale)


# ---- the three emission paths of the hinted forward (csrc/raster_api.hip): order -> record gathers (R2_SORTED_RECORDS=0), sorted
# records (R2_EMIT_HIST=0), and emission by output range fused with the tile sort's histograms (default) -- identical lists
_BIN_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from r2_gaussian_amd import scene as S
from tests import helpers as Hh
dev = torch.device("cuda:0")
out = {}
for tag, P, det, seed in (("a", 30000, (160, 144), 21), ("b", 120000, (512, 512), 5), ("c", 7, (48, 48), 3)):
    c = S.make_cloud(P, seed=seed)
    views = S.make_views(8, det)
    Hh.hip_raster(c, views[2], dev)          # un-hinted first call for this size
    h = Hh.hip_raster(c, views[3], dev)      # hinted
    assert h["host_words"][7] > 0            # DW_NVIS: the hinted path ran
    for k in ("point_list", "ranges", "first", "tiles_unsorted", "vals_unsorted", "offsets", "order", "color"):
        out[tag + "_" + k] = h[k] if k != "order" and k != "offsets" else h[k][:h["host_words"][7]]
    out[tag + "_first"] = h["first"] * (h["radii"] > 0)   # (rows of culled Gaussians are never written)
    out[tag + "_R"] = np.array([h["num_rendered"]])
np.savez(sys.argv[1], **out)
"""


def _run_env(tmp_path, name, extra_env):
    out = str(tmp_path / (name + ".npz"))
    env = dict(os.environ)
    env.pop("R2HIP_LIB", None)
    env.update(extra_env)
    r = subprocess.run([sys.executable, "-c", _BIN_SCRIPT % ROOT, out], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out)


def test_emission_paths_produce_identical_lists(tmp_path):
    off = {"R2_TILE_FIRST": "0"}   # the three emission kernels belong to the general chain
    ref = _run_env(tmp_path, "gathers", dict(off, R2_SORTED_RECORDS="0"))
    for name, env in (("sorted_records", dict(off, R2_EMIT_HIST="0")), ("fused", off)):
        got = _run_env(tmp_path, name, env)
        for k in ref.files:
            assert np.array_equal(ref[k], got[k]), (name, k)
    # ... and the tile-first chain (round 4), which has no emission list at all: what the reference defines -- point_list,
    # ranges, num_rendered -- and the image are bit-identical
    got = _run_env(tmp_path, "tile_first", {})
    for k in ref.files:
        if k.split("_", 1)[1] in ("point_list", "ranges", "color", "R"):
            assert np.array_equal(ref[k], got[k]), ("tile_first", k)
