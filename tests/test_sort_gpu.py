"""The hand-written stable radix sort (radix_sort.hip) through the binning pipeline's real entry point is
covered by the parity tests; this file hammers it directly on adversarial key distributions via a tiny
ctypes hook-free route: the voxelizer/rasterizer sorts are exercised with sizes that hit every
items-per-thread variant and ragged tails."""
import numpy as np
import pytest
import torch

from r2_gaussian_amd import scene as S
from tests import helpers as Hh

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P,hw", [(1, 32), (63, 48), (257, 64), (1025, 64), (140000, 256), (600000, 512), (30000, 1040)])
def test_sorted_lists_match_oracle_across_sizes(P, hw, oracle, gpu):
    """P / R span one to hundreds of 4096-key sort tiles with ragged tails; 1040^2 has 4225 tiles = 13 key bits, i.e. the
    two-pass tile sort + boundary-scan ranges instead of the single-pass sort whose digit totals give the ranges."""
    c = S.make_cloud(P, seed=P % 13, scale_mult=1.3 if P > 1000 else 3.0)
    v = S.make_view(0.1 * (P % 7), (hw, hw))
    o = Hh.oracle_raster(oracle, c, v, render=False)
    h = Hh.hip_raster(c, v, gpu)
    Hh.check_binning(h, o)


def test_many_equal_depths_are_stable(oracle, gpu):
    """All Gaussians at the same depth (identical low sort word): order inside a tile must be ascending id."""
    P = 5000
    c = S.make_cloud(P, seed=3)
    v = S.make_view(0.0, (128, 128))
    # view 0 looks along -x from (5,0,0): depth = 5 - x.  Pin x so that every depth key is identical.
    xyz = c.xyz.clone()
    xyz[:, 0] = 0.25
    cl = S.Cloud(xyz, c.scales, c.rotations, c.density)
    o = Hh.oracle_raster(oracle, cl, v, render=False)
    assert len(np.unique(o["depths"][o["radii"] > 0])) == 1
    h = Hh.hip_raster(cl, v, gpu)
    Hh.check_binning(h, o)
    rg = h["ranges"]
    for t in np.nonzero(rg[:, 1] > rg[:, 0])[0][:50]:
        seg = h["point_list"][rg[t, 0]:rg[t, 1]]
        assert (np.diff(seg.astype(np.int64)) > 0).all()


def test_depth_hint_paths_are_identical_and_stale_hints_are_harmless(oracle, gpu):
    """The forward orders the Gaussians by depth with buckets shaped by the PREVIOUS call's depth range (a hint).  The
    hinted and the un-hinted path must give bit-identical binning, and a hint that no longer fits the scene (depths
    compressed into a few buckets, or entirely outside the hinted range) must only cost a fallback, never exactness."""
    from r2_gaussian_amd import _lib
    L = _lib.lib()
    P = 20000
    c = S.make_cloud(P, seed=5)
    v = S.make_view(0.0, (256, 256))   # looks along -x from (5, 0, 0): depth = 5 - x
    try:
        L.r2_tile_first_control(0)                     # this test is about the general chain's depth order (tests/test_tilefirst_gpu.py has the other)
        L.r2_depth_hint_control(2)                     # forget the history: first call is un-hinted
        h0 = Hh.hip_raster(c, v, gpu)
        assert int(h0["host_words"][7]) == 0
        h1 = Hh.hip_raster(c, v, gpu)                  # same P again: hinted
        assert int(h1["host_words"][7]) == int((h1["tiles_touched"] > 0).sum()) > 0
        assert int(h1["host_words"][1]) == 0
        for k in ("tiles_touched", "tiles_unsorted", "vals_unsorted", "point_list", "ranges", "first", "color"):
            assert np.array_equal(h0[k], h1[k]), k
        o = Hh.oracle_raster(oracle, c, v, render=False)
        Hh.check_binning(h1, o)

        def shifted(scale, dx):
            xyz = c.xyz.clone()
            xyz[:, 0] = xyz[:, 0] * scale + dx
            return S.Cloud(xyz, c.scales, c.rotations, c.density)

        # depths squeezed into ~1% of the hinted range: fuller buckets, still the fast path
        c2 = shifted(0.01, 0.0)
        h2 = Hh.hip_raster(c2, v, gpu)
        Hh.check_binning(h2, Hh.oracle_raster(oracle, c2, v, render=False))
        L.r2_depth_hint_control(2)
        Hh.hip_raster(c, v, gpu)                       # re-arm the wide hint
        # ... into a handful of buckets: overflow -> radix fallback
        c3 = shifted(1e-5, 0.0)
        h3 = Hh.hip_raster(c3, v, gpu)
        assert int(h3["host_words"][1]) == 1
        Hh.check_binning(h3, Hh.oracle_raster(oracle, c3, v, render=False))
        L.r2_depth_hint_control(2)
        Hh.hip_raster(c, v, gpu)
        # ... entirely outside the hinted range (everything clamps into the first bucket)
        c4 = shifted(1.0, 3.0)
        h4 = Hh.hip_raster(c4, v, gpu)
        assert int(h4["host_words"][1]) == 1
        Hh.check_binning(h4, Hh.oracle_raster(oracle, c4, v, render=False))
        # the refreshed hint serves the next call of the shifted scene
        h5 = Hh.hip_raster(c4, v, gpu)
        assert int(h5["host_words"][1]) == 0 and int(h5["host_words"][7]) > 0
        assert np.array_equal(h5["point_list"], h4["point_list"]) and np.array_equal(h5["color"], h4["color"])
        # part of the cloud behind the near plane (depth = 0.5 - x <= 0.2 is culled): the hinted path only writes the
        # visible prefix of the depth order, the culled Gaussians must stay out of every list
        c7 = shifted(1.0, 4.5)
        o7 = Hh.oracle_raster(oracle, c7, v, render=False)
        nvis7 = int((o7["tiles_touched"] > 0).sum())
        assert 0 < nvis7 < P
        Hh.hip_raster(c7, v, gpu)                      # stale hint (overflow or not), then a fresh one
        h7 = Hh.hip_raster(c7, v, gpu)
        assert int(h7["host_words"][1]) == 0 and int(h7["host_words"][7]) == nvis7
        Hh.check_binning(h7, o7)
        # hints switched off: always the un-hinted path
        L.r2_depth_hint_control(0)
        h6 = Hh.hip_raster(c4, v, gpu)
        assert int(h6["host_words"][7]) == 0 and np.array_equal(h6["point_list"], h4["point_list"])
    finally:
        L.r2_depth_hint_control(1)
        L.r2_depth_hint_control(2)
        L.r2_tile_first_control(1)
