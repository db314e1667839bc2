"""Stick-first binning of the voxelizer (csrc/voxel_sticks.hip; grids of more than 4096 tiles, e.g. the 256^3 query of
test.py:105-112): the lists it builds are the reference's bit for bit (oracle), volumes and gradients are identical to the general
chain's, and a scene it cannot serve continues on the general chain."""
import ctypes as C

import numpy as np
import pytest
import torch

from r2_gaussian_amd import scene as S
from tests import helpers as Hh

pytestmark = pytest.mark.gpu


def _lib():
    from r2_gaussian_amd import _lib as L
    return L.lib()


def _stats(reset=True):
    st = (C.c_longlong * 3)()
    _lib().r2_voxel_sticks_stats(st, 1 if reset else 0)
    return list(st)   # taken, fallback, declined


@pytest.fixture
def sticks_mode():
    """-> a setter of the chain's mode (0 off, 1 on); the default and the thread's notes are restored afterwards"""
    L = _lib()
    L.r2_voxel_sticks_control(3)
    yield L.r2_voxel_sticks_control
    L.r2_voxel_sticks_control(1)
    L.r2_voxel_sticks_control(5)
    L.r2_voxel_sticks_control(3)


# (P, nVoxel, sVoxel, center, scale_mult, mode)
CASES = [
    (5000, (64, 64, 64), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0), 1.0, 1),            # 512 tiles: a list is a tile (shift 0)
    (3000, (40, 28, 52), (2.0, 1.4, 2.6), (0.05, 0.0, -0.1), 1.5, 1),          # ragged grid, 140 tiles
    (4000, (136, 136, 136), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0), 1.0, 1),         # 4913 tiles: sticks of 2, the last one partial
    (6000, (168, 136, 104), (2.0, 1.6, 1.2), (0.02, -0.03, 0.3), 1.0, 1),      # 4641 tiles, ragged, off-centre in z
    (20000, (256, 256, 256), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0), 1.0, 1),        # 32768 tiles: sticks of 8 (the headline query's grid)
]
IDS = ["64cube_shift0", "ragged_40x28x52", "136cube_shift1", "ragged_168x136x104", "256cube_shift3"]


def _cloud(case):
    return S.make_cloud(case[0], seed=case[0] % 89, scale_mult=case[4])


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_lists_are_the_oracles_bit_for_bit(case, oracle, gpu, sticks_mode):
    P, n, s, ctr, sm, mode = case
    c = _cloud(case)
    o = Hh.oracle_voxel(oracle, c, n, s, ctr, render=False)
    sticks_mode(mode)
    _stats()
    h = Hh.hip_voxel(c, n, s, ctr, gpu)
    assert Hh.took_sticks(h) and _stats()[0] == 1
    assert h["num_rendered"] == o["num_rendered"] > 0
    for k in ("radii_x", "radii_y", "radii_z"):
        assert np.array_equal(h[k], o[k])
    Hh.check_binning(h, o)   # sorted (tile | z bits) keys, point_list, ranges; the backward's rows partition [0, R)
    assert (c.xyz[:, 2] < 0).any() and (c.xyz[:, 2] > 0).any()   # lists straddle z = 0: both sign classes of the key (quirk Q10)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_volume_and_gradients_are_those_of_the_general_chain(case, gpu, sticks_mode):
    """Same lists, same render kernels: the volume is identical bit for bit; the backward's moment rows sit elsewhere (first
    instance in id order instead of depth order) but every Gaussian's rows are summed in the same order: identical gradients."""
    P, n, s, ctr, sm, mode = case
    c = _cloud(case)
    g = torch.Generator().manual_seed(1)
    dL = ((torch.rand(*n, generator=g) * 2 - 1) / float(np.prod(n))).numpy()
    sticks_mode(0)
    h0 = Hh.hip_voxel(c, n, s, ctr, gpu)
    g0 = Hh.hip_voxel_backward(h0, c, n, s, ctr, dL, gpu)
    sticks_mode(mode)
    h1 = Hh.hip_voxel(c, n, s, ctr, gpu)
    g1 = Hh.hip_voxel_backward(h1, c, n, s, ctr, dL, gpu)
    assert not Hh.took_sticks(h0) and Hh.took_sticks(h1)
    assert h0["num_rendered"] == h1["num_rendered"]
    assert np.array_equal(h0["point_list"], h1["point_list"]) and np.array_equal(h0["ranges"], h1["ranges"])
    assert np.array_equal(h0["vol"].view(np.uint32), h1["vol"].view(np.uint32))
    assert h1["vol"].max() > 0
    for k in g0:
        assert np.array_equal(g0[k].view(np.uint32), g1[k].view(np.uint32)), k
    assert np.abs(g1["dL_dmeans3D"]).max() > 0


def test_volume_within_1e4_of_the_oracle(oracle, gpu, sticks_mode):
    case = CASES[2]
    P, n, s, ctr, sm, mode = case
    c = _cloud(case)
    o = Hh.oracle_voxel(oracle, c, n, s, ctr)
    sticks_mode(mode)
    h = Hh.hip_voxel(c, n, s, ctr, gpu)
    assert Hh.took_sticks(h)
    st = Hh.parity_volume(oracle, o, h["vol"], "voxel sticks P=%d %s" % (P, "x".join(map(str, n))))
    assert st["n_flip_candidates"] < 0.01 * st["n"]
    assert o["vol"].max() > 0.01


def test_the_headline_query_takes_the_chain(gpu, sticks_mode):
    """300k Gaussians, 256^3 (BASELINE's voxelizer workload): default mode, lists and volume identical to the general chain's."""
    c = S.make_cloud(300000, seed=0)
    n, s, ctr = (256, 256, 256), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0)
    sticks_mode(0)
    h0 = Hh.hip_voxel(c, n, s, ctr, gpu)
    sticks_mode(1)
    _stats()
    h1 = Hh.hip_voxel(c, n, s, ctr, gpu)
    assert _stats() == [1, 0, 0] and Hh.took_sticks(h1) and not Hh.took_sticks(h0)
    assert h0["num_rendered"] == h1["num_rendered"] > 4000000
    assert np.array_equal(h0["point_list"], h1["point_list"]) and np.array_equal(h0["ranges"], h1["ranges"])
    assert np.array_equal(h0["vol"].view(np.uint32), h1["vol"].view(np.uint32))
    keys = h1["keys"]
    assert (keys[1:] >= keys[:-1]).all()
    tt = h1["tiles_touched"].astype(np.int64)
    vis = np.nonzero(tt > 0)[0]
    start = h1["first"].astype(np.int64)[vis]
    assert np.array_equal(np.cumsum(tt[vis]) - tt[vis], start)   # moment rows in id order, a partition of [0, R)


def _squeezed(P, seed, f, flat_z=False, scale_mult=1.0):
    c0 = S.make_cloud(P, seed=seed, scale_mult=scale_mult)
    xyz = c0.xyz * f
    if flat_z:
        xyz = xyz.clone()
        xyz[:, 2] = 0.0123
    return S.Cloud(xyz, c0.scales, c0.rotations, c0.density)


LONG = [
    # (cloud, grid): every Gaussian squeezed into a few tiles -- lists of 12 000 .. 60 000 instances, sorted by several workgroups each
    ("tile_lists_64cube", lambda: _squeezed(20000, 5, 0.03), (64, 64, 64)),            # a list is a tile: parts are ranges of z
    ("stick_lists_136cube", lambda: _squeezed(30000, 7, 0.05), (136, 136, 136)),       # sticks of 2 tiles: parts are ranges of (tile, z)
    ("stick_lists_256cube", lambda: _squeezed(40000, 9, 0.1, scale_mult=0.5), (256, 256, 256)),   # sticks of 8
    ("equal_z_64cube", lambda: _squeezed(12000, 11, 0.03, flat_z=True), (64, 64, 64)),  # one value of z: a part cannot be cut -> ranked by counting
]


@pytest.mark.parametrize("name,make,n", LONG, ids=[c[0] for c in LONG])
def test_lists_beyond_one_workgroup_are_sorted_in_parts(name, make, n, oracle, gpu, sticks_mode):
    c = make()
    s, ctr = (2.0, 2.0, 2.0), (0.0, 0.0, 0.0)
    o = Hh.oracle_voxel(oracle, c, n, s, ctr, render=False)
    assert int((o["ranges"][:, 1].astype(np.int64) - o["ranges"][:, 0]).max()) > 8192
    sticks_mode(0)
    h0 = Hh.hip_voxel(c, n, s, ctr, gpu)
    sticks_mode(1)
    _stats()
    h = Hh.hip_voxel(c, n, s, ctr, gpu)
    assert _stats() == [1, 0, 0] and Hh.took_sticks(h) and int(h["host_words"][3]) > 8192   # (host word 3: the longest list)
    assert h["num_rendered"] == o["num_rendered"]
    Hh.check_binning(h, o)
    assert np.array_equal(h0["vol"].view(np.uint32), h["vol"].view(np.uint32))
    g = torch.Generator().manual_seed(2)
    dL = ((torch.rand(*n, generator=g) * 2 - 1) / float(np.prod(n))).numpy()
    g0 = Hh.hip_voxel_backward(h0, c, n, s, ctr, dL, gpu)
    g1 = Hh.hip_voxel_backward(h, c, n, s, ctr, dL, gpu)
    for k in g0:
        assert np.array_equal(g0[k].view(np.uint32), g1[k].view(np.uint32)), k


def test_gaussians_of_thousands_of_tiles_are_walked_by_their_wave(oracle, gpu, sticks_mode):
    """A trained scene holds a few Gaussians that span much of the volume (background blobs): their tile cubes -- here 2197 and 4913
    tiles on a 136^3 grid, among 5000 ordinary Gaussians -- are walked by the whole wave, a row per lane, in the count and scatter
    kernels (one lane walked for 235 us on the 92k trained cloud)."""
    c0 = S.make_cloud(5000, seed=4)
    scales = c0.scales.clone()
    scales[7] = 0.25       # radius 3 sigma = 0.75 of a 2.0 volume: a cube of 13 tiles a side
    scales[1500] = 0.6     # covers the whole grid
    scales[4999] = 0.6
    c = S.Cloud(c0.xyz, scales, c0.rotations, c0.density)
    n, s, ctr = (136, 136, 136), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0)
    o = Hh.oracle_voxel(oracle, c, n, s, ctr, render=False)
    assert int(o["tiles_touched"].max()) == 17 ** 3 and int((o["tiles_touched"] > 256).sum()) >= 3
    sticks_mode(0)
    h0 = Hh.hip_voxel(c, n, s, ctr, gpu)
    sticks_mode(1)
    _stats()
    h = Hh.hip_voxel(c, n, s, ctr, gpu)
    assert _stats() == [1, 0, 0] and Hh.took_sticks(h)
    assert h["num_rendered"] == o["num_rendered"]
    Hh.check_binning(h, o)
    assert np.array_equal(h0["vol"].view(np.uint32), h["vol"].view(np.uint32))
    g = torch.Generator().manual_seed(3)
    dL = ((torch.rand(*n, generator=g) * 2 - 1) / float(np.prod(n))).numpy()
    g0 = Hh.hip_voxel_backward(h0, c, n, s, ctr, dL, gpu)
    g1 = Hh.hip_voxel_backward(h, c, n, s, ctr, dL, gpu)
    for k in g0:
        assert np.array_equal(g0[k].view(np.uint32), g1[k].view(np.uint32)), k


def test_a_large_scene_with_very_long_lists_continues_on_the_general_chain(oracle, gpu, sticks_mode):
    """A list of more than 20 480 instances in a scene of more than 8 Mi (the 331k trained cloud: lists of 28 000, 22 M instances): the
    part-wise sort costs more than the general chain's radix passes, the stick chain hands over after the preprocess.  The two
    limits are lowered here so that a small scene shows the rule."""
    c = _squeezed(20000, 5, 0.03)
    n, s, ctr = (64, 64, 64), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0)
    o = Hh.oracle_voxel(oracle, c, n, s, ctr, render=False)
    L = _lib()
    try:
        L.r2_voxel_sticks_limits(10000, 100000)     # the scene: lists of 20 000, 160 000 instances
        sticks_mode(1)
        _stats()
        h = Hh.hip_voxel(c, n, s, ctr, gpu)
        assert _stats() == [0, 1, 0] and not Hh.took_sticks(h)
        assert h["num_rendered"] == o["num_rendered"]
        Hh.check_binning(h, o)
        h2 = Hh.hip_voxel(c, n, s, ctr, gpu)                      # the thread remembers: declined, the hinted general chain
        assert _stats() == [0, 0, 1]
        assert np.array_equal(h2["point_list"], h["point_list"]) and np.array_equal(h2["vol"].view(np.uint32), h["vol"].view(np.uint32))
        L.r2_voxel_sticks_limits(10000, 1000000)    # only one of the two limits exceeded: the chain serves it
        sticks_mode(3)
        h3 = Hh.hip_voxel(c, n, s, ctr, gpu)
        assert _stats() == [1, 0, 0] and Hh.took_sticks(h3)
    finally:
        L.r2_voxel_sticks_limits(0, 0)


def test_a_scene_the_chain_cannot_serve_continues_on_the_general_one(oracle, gpu, sticks_mode):
    """The chain gives up after its scan when the long lists of a scene need more part descriptors than the geometry state holds
    (Gaussians of hundreds of tiles each); forced here by declaring lists beyond one workgroup's capacity unsupported (mode 4).
    The call finishes on the general chain (same result, the preprocess is not repeated), and the thread skips the chain for that
    (P, grid) from then on."""
    c = _squeezed(20000, 5, 0.03)
    n, s, ctr = (64, 64, 64), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0)
    o = Hh.oracle_voxel(oracle, c, n, s, ctr, render=False)
    sticks_mode(1)
    sticks_mode(4)
    _stats()
    h = Hh.hip_voxel(c, n, s, ctr, gpu)
    assert _stats() == [0, 1, 0] and not Hh.took_sticks(h)
    assert h["num_rendered"] == o["num_rendered"]
    Hh.check_binning(h, o)
    h2 = Hh.hip_voxel(c, n, s, ctr, gpu)
    assert _stats() == [0, 0, 1] and not Hh.took_sticks(h2)
    assert np.array_equal(h2["point_list"], h["point_list"]) and np.array_equal(h2["vol"].view(np.uint32), h["vol"].view(np.uint32))
    sticks_mode(3)   # notes forgotten: it tries again
    Hh.hip_voxel(c, n, s, ctr, gpu)
    assert _stats() == [0, 1, 0]


def test_nothing_visible(gpu, sticks_mode):
    c = S.make_cloud(400, seed=2)
    n, s, ctr = (136, 136, 136), (0.1, 0.1, 0.1), (30.0, 30.0, 30.0)   # volume far from every Gaussian
    sticks_mode(1)
    h = Hh.hip_voxel(c, n, s, ctr, gpu)
    assert Hh.took_sticks(h) and h["num_rendered"] == 0 and not h["vol"].any() and not h["ranges"].any()
