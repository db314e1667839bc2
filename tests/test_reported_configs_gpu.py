"""GPU parity AT THE CONFIGURATIONS THAT ARE REPORTED (bench.py / BASELINE.json), not only at small ones:

* the headline workload itself -- 300k Gaussians (seed 0), 512^2 cone-beam detector, views 0 and 17 of the 50-view set:
  bit-exact binning (radii, depth order, tile runs, sorted (tile|depth) keys, point_list, ranges), image within the pure 1e-4
  relative bound + attributed cut-off flips, the full backward against the oracle's double sums;
* the voxelizer's reported query -- 300k Gaussians on 256^3: bit-exact indices, volume parity; and the training loop's 32^3 TV
  patch at 300k (forward + backward), the configuration bench.py times as `tv_patch_32cube_fwd_bwd_us`;
* BASELINE config E -- 1M Gaussians, 1024^2 detector (T = 4096 tiles: the single-pass 12-bit tile sort): bit-exact indices,
  image parity, full backward.

The oracle needs seconds per case on the GPU box's host cores (OpenMP).
"""
import numpy as np
import pytest
import torch

from r2_gaussian_amd import scene as S
from tests import helpers as Hh

pytestmark = pytest.mark.gpu


def _raster_full(oracle, gpu, c, v, label, backward=True, max_candidate_frac=0.01):
    o = Hh.oracle_raster(oracle, c, v)
    h = Hh.hip_raster(c, v, gpu)
    assert h["num_rendered"] == o["num_rendered"] > 0
    assert np.array_equal(h["radii"], o["radii"])
    Hh.check_binning(h, o)
    st = Hh.parity_image(oracle, o, h["color"], label)
    assert st["n_flip_candidates"] < max_candidate_frac * st["n"]
    if backward:
        dL = S.make_pixel_grad(v.image_height, v.image_width).numpy()
        gh = Hh.hip_raster_backward(h, c, v, dL, gpu)
        sg = Hh.parity_raster_grads(oracle, o, gh, c, v, dL, label)
        for k in ("dL_dmeans2D", "dL_dopacity", "dL_dmu", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations"):
            assert sg[k]["max_err_over_scale_unflagged"] <= 2e-4, (k, sg[k])
    return o, h


@pytest.mark.parametrize("view", [0, 17])
def test_headline_300k_512_forward_backward(view, oracle, gpu):
    c = S.make_cloud(300000, seed=0)                   # bench.py's cloud
    v = S.make_views(50, (512, 512))[view]             # ... and its view set
    o, _h = _raster_full(oracle, gpu, c, v, "HEADLINE 300k/512^2 view %d" % view)
    assert 0.8e6 < o["num_rendered"] < 1.6e6


def test_config_E_1M_1024(oracle, gpu):
    c = S.make_cloud(1000000, seed=0)
    v = S.make_views(360, (1024, 1024))[41]
    o, h = _raster_full(oracle, gpu, c, v, "config E 1M/1024^2")
    assert o["ranges"].shape[0] == 4096


def test_voxelizer_300k_256cube(oracle, gpu):
    c = S.make_cloud(300000, seed=0)
    n, s, ctr = (256, 256, 256), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0)   # bench.py's query
    o = Hh.oracle_voxel(oracle, c, n, s, ctr)
    h = Hh.hip_voxel(c, n, s, ctr, gpu)
    assert h["num_rendered"] == o["num_rendered"] > 4e6
    for k in ("radii_x", "radii_y", "radii_z"):
        assert np.array_equal(h[k], o[k]), k
    Hh.check_binning(h, o)
    st = Hh.parity_volume(oracle, o, h["vol"], "voxelizer 300k/256^3")
    assert st["n_flip_candidates"] < 0.01 * st["n"]


def test_tv_patch_300k_32cube_forward_backward(oracle, gpu):
    c = S.make_cloud(300000, seed=0)
    n, s, ctr = (32, 32, 32), (0.25, 0.25, 0.25), (-0.1, 0.0, 0.05)   # one of bench.py's TV patches (train.py:128-142)
    o = Hh.oracle_voxel(oracle, c, n, s, ctr)
    h = Hh.hip_voxel(c, n, s, ctr, gpu)
    assert h["num_rendered"] == o["num_rendered"] > 0
    Hh.check_binning(h, o)
    Hh.parity_volume(oracle, o, h["vol"], "TV patch 300k/32^3")
    g = torch.Generator().manual_seed(1)
    dL = ((torch.rand(*n, generator=g) * 2 - 1) / float(np.prod(n))).numpy()
    gh = Hh.hip_voxel_backward(h, c, n, s, ctr, dL, gpu)
    sg = Hh.parity_voxel_grads(oracle, o, gh, c, dL, "TV patch 300k/32^3")
    for k in ("dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations"):
        assert sg[k]["max_err_over_scale_unflagged"] <= 2e-4, (k, sg[k])


# ---- gradient parity beyond one cloud: other seeds, half / double the Gaussian size, scale_modifier 1.5.
# Every case logs its margins (max_err_over_tol per gradient) to gpurun_out/parity_report.json; the committed copy is
# profiles/r03_parity_report.json.
@pytest.mark.parametrize("scale_modifier", [1.0, 1.5], ids=["mod1", "mod1.5"])
@pytest.mark.parametrize("scale_mult", [0.5, 1.0, 2.0], ids=["half", "unit", "double"])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_gradient_parity_sweep_300k_512(seed, scale_mult, scale_modifier, oracle, gpu):
    if seed == 0 and scale_mult == 1.0 and scale_modifier == 1.0:
        pytest.skip("the headline case above")
    c = S.make_cloud(300000, seed=seed, scale_mult=scale_mult)
    v = S.make_views(50, (512, 512))[(7 * seed + 3) % 50]
    label = "SWEEP 300k/512^2 seed %d scale x%g modifier %g" % (seed, scale_mult, scale_modifier)
    o = Hh.oracle_raster(oracle, c, v, scale_modifier=scale_modifier)
    h = Hh.hip_raster(c, v, gpu, scale_modifier=scale_modifier)
    assert h["num_rendered"] == o["num_rendered"] > 0
    assert np.array_equal(h["radii"], o["radii"])
    Hh.check_binning(h, o)
    Hh.parity_image(oracle, o, h["color"], label)
    dL = S.make_pixel_grad(512, 512, seed=seed + 1).numpy()
    gh = Hh.hip_raster_backward(h, c, v, dL, gpu)
    sg = Hh.parity_raster_grads(oracle, o, gh, c, v, dL, label, scale_modifier=scale_modifier)
    for k in ("dL_dmeans2D", "dL_dopacity", "dL_dmu", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations"):
        assert sg[k]["max_err_over_scale_unflagged"] <= 2e-4, (k, sg[k])


def test_config_C_300k_560(oracle, gpu):
    """BASELINE config C (pine-like): ~300k Gaussians, 560^2 detector (data_generator/real_dataset/README.md:68) -- T = 35 x 35
    = 1225 tiles, 11 tile-id bits."""
    c = S.make_cloud(300000, seed=3)
    v = S.make_views(50, (560, 560))[11]
    o, _h = _raster_full(oracle, gpu, c, v, "config C 300k/560^2")
    assert o["ranges"].shape[0] == 1225


# ---- TRAINED, DENSIFIED clouds (BASELINE configs[2] "full densification to ~300k Gaussians").  Everything above
# draws its Gaussians from scene.make_cloud (uniform in an ellipsoid, log-uniform scales).  These clouds went through the
# training loop's clone / split / prune rounds (train.py:155-168, gaussian_model.py:503-550) on the synthetic cone-beam case at
# 512^2 / 256^3: 50k -> 92k ("small", round 2/3's run) and 50k -> ~335k ("large", lowered gradient threshold, capped at 300k),
# saved and re-loaded in the reference's point_cloud.pickle layout (tests/trained_cloud.py trains them on this GPU when no copy
# is around).  They are a different regime: ~9.3 instances per Gaussian instead of 3.9, tile lists of up to 23k entries,
# depth keys clustered on the object.
@pytest.fixture(scope="module", params=["small", "large"])
def trained(request, gpu):
    from tests import trained_cloud as TC
    c, info = TC.load(request.param)
    assert c is not None
    return request.param, c, info


@pytest.mark.parametrize("view", [0, 17])
def test_trained_cloud_raster_forward_backward(trained, view, oracle, gpu):
    name, c, info = trained
    P = c.xyz.shape[0]
    v = S.make_views(50, (512, 512))[view]
    label = "TRAINED %s (P %d) 512^2 view %d" % (name, P, view)
    Hh.hip_raster(c, v, gpu)                       # first call with this P: un-hinted; the one checked below is the hinted path
    # ~3000 Gaussians per pixel on the large cloud (R x 256 / N) against ~1100 on the synthetic one: proportionally more pixels
    # hold a pair that sits on the cut-off (2.5 % at 335k); how many of them NEED their budget is in the report (0 - 3)
    o, h = _raster_full(oracle, gpu, c, v, label, max_candidate_frac=0.05)
    assert int(h["host_words"][7]) == int((o["tiles_touched"] > 0).sum()), "the hinted depth order did not run"
    assert int(h["host_words"][1]) == 0, "the hinted depth order overflowed its buckets on a trained cloud"
    assert o["num_rendered"] > 6 * P               # densified clouds are instance-heavy: the regime the synthetic cloud lacks
    L = (o["ranges"][:, 1] - o["ranges"][:, 0]).astype(np.int64)
    Hh.PARITY_LOG.append(dict(kind="cloud_stats", case=label, P=int(P), R=int(o["num_rendered"]),
                              list_len_p50=float(np.percentile(L, 50)), list_len_p90=float(np.percentile(L, 90)),
                              list_len_p99=float(np.percentile(L, 99)), list_len_max=int(L.max()),
                              hint_overflow=int(h["host_words"][1]), thin_flag=int(h["host_words"][2])))


def test_trained_cloud_query_256cube_and_tv_patch(trained, oracle, gpu):
    name, c, info = trained
    P = c.xyz.shape[0]
    n, s, ctr = (256, 256, 256), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0)
    o = Hh.oracle_voxel(oracle, c, n, s, ctr)
    Hh.hip_voxel(c, n, s, ctr, gpu)
    h = Hh.hip_voxel(c, n, s, ctr, gpu)            # second call: hinted
    assert h["num_rendered"] == o["num_rendered"] > 0
    for k in ("radii_x", "radii_y", "radii_z"):
        assert np.array_equal(h[k], o[k]), k
    Hh.check_binning(h, o)
    Hh.parity_volume(oracle, o, h["vol"], "TRAINED %s query 256^3" % name)
    # one TV patch inside the object (train.py:128-142), forward + backward
    n, s, ctr = (32, 32, 32), (0.25, 0.25, 0.25), (-0.1, 0.0, 0.05)
    o = Hh.oracle_voxel(oracle, c, n, s, ctr)
    h = Hh.hip_voxel(c, n, s, ctr, gpu)
    assert h["num_rendered"] == o["num_rendered"] > 0
    Hh.check_binning(h, o)
    Hh.parity_volume(oracle, o, h["vol"], "TRAINED %s TV patch 32^3" % name)
    g = torch.Generator().manual_seed(1)
    dL = ((torch.rand(*n, generator=g) * 2 - 1) / float(np.prod(n))).numpy()
    gh = Hh.hip_voxel_backward(h, c, n, s, ctr, dL, gpu)
    sg = Hh.parity_voxel_grads(oracle, o, gh, c, dL, "TRAINED %s TV patch 32^3" % name)
    for k in ("dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations"):
        assert sg[k]["max_err_over_scale_unflagged"] <= 2e-4, (k, sg[k])
