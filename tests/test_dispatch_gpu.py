"""The dispatch table (csrc/dispatch.hpp, r2_path_stats): which chain a forward takes and WHY the others did not, through the public
knobs only.  Results never depend on the chain (every other GPU test file compares them); this one checks the bookkeeping."""
import numpy as np
import pytest
import torch

from r2_gaussian_amd import scene as S
from tests import helpers as Hh

pytestmark = pytest.mark.gpu


def _delta(before, after):
    return {k: after[k] - before.get(k, 0) for k in after if after[k] != before.get(k, 0)}


def test_rasterizer_chains_and_reasons(gpu):
    from r2_gaussian_amd import _lib
    L = _lib.lib()
    L.r2_tile_first_control(1)
    L.r2_tile_first_control(2)     # forget this thread's predictions
    c = S.make_cloud(7000, seed=3)
    v = S.make_views(4, (160, 160))[1]
    b = _lib.path_stats()
    Hh.hip_raster(c, v, gpu)                        # nothing to predict from
    Hh.hip_raster(c, v, gpu)                        # predicted
    Hh.hip_raster(c, v, gpu, debug=True)            # debug mode
    L.r2_tile_first_control(0)
    Hh.hip_raster(c, v, gpu)                        # switched off
    L.r2_tile_first_control(1)
    c2 = S.make_cloud(9000, seed=3)
    Hh.hip_raster(c2, v, gpu)                       # a new Gaussian count on a known detector: seeded
    d = _delta(b, _lib.path_stats())
    assert d == {"raster.general.no_prediction": 1, "raster.tile_first": 2, "raster.general.debug": 1,
                 "raster.general.switched_off": 1, "raster.event.seeded": 1}, d
    # a detector of more than 4096 tiles: the general chain, and it says why
    b = _lib.path_stats()
    big = S.make_views(2, (1040, 1040))[0]
    Hh.hip_raster(S.make_cloud(2000, seed=1), big, gpu)
    assert _delta(b, _lib.path_stats()) == {"raster.general.grid": 1}


def test_voxelizer_chains_and_reasons(gpu):
    from r2_gaussian_amd import _lib
    L = _lib.lib()
    L.r2_voxel_sticks_control(1)
    L.r2_voxel_sticks_control(3)
    c = S.make_cloud(20000, seed=5)
    b = _lib.path_stats()
    Hh.hip_voxel(c, (32, 32, 32), (0.25, 0.25, 0.25), (0.1, -0.05, 0.2), gpu)              # 64 tiles: the patch path
    Hh.hip_voxel(c, (64, 64, 64), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0), gpu)                   # 512 tiles: stick-first
    Hh.hip_voxel(c, (64, 64, 64), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0), gpu, debug=True)       # debug
    Hh.hip_voxel(c, (32, 32, 32), (0.25, 0.25, 0.25), (0.1, -0.05, 0.2), gpu, slab=(1, 3))  # an x-slab of 32 tiles
    L.r2_voxel_sticks_control(0)
    Hh.hip_voxel(c, (64, 64, 64), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0), gpu)                   # switched off
    L.r2_voxel_sticks_control(1)
    d = _delta(b, _lib.path_stats())
    assert d == {"voxel.small_grid": 1, "voxel.stick_first": 1, "voxel.general.debug": 1, "voxel.general.slab": 1,
                 "voxel.general.switched_off": 1}, d
    # the long-list rule: handed over after the scan, remembered, tried again after the limits change
    L.r2_voxel_sticks_limits(1, 1)
    try:
        b = _lib.path_stats()
        h1 = Hh.hip_voxel(c, (64, 64, 64), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0), gpu)
        h2 = Hh.hip_voxel(c, (64, 64, 64), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0), gpu)
        assert _delta(b, _lib.path_stats()) == {"voxel.general.long_lists": 1, "voxel.general.remembered": 1}
    finally:
        L.r2_voxel_sticks_limits(0, 0)
    b = _lib.path_stats()
    h3 = Hh.hip_voxel(c, (64, 64, 64), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0), gpu)             # the notes went with the old limits
    assert _delta(b, _lib.path_stats()) == {"voxel.stick_first": 1}
    for h in (h1, h2):
        assert np.array_equal(h["vol"].view(np.uint32), h3["vol"].view(np.uint32)) and np.array_equal(h["point_list"], h3["point_list"])
