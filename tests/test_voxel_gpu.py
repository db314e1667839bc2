"""GPU parity of the voxelizer (HIP via the C ABI) vs the CPU oracle: bit-exact tile / sort indices,
volumes within 1e-4 relative, gradients against double-accumulated oracle sums (incl. reference quirk Q4)."""
import numpy as np
import pytest
import torch

from r2_gaussian_amd import scene as S
from tests import helpers as Hh

pytestmark = pytest.mark.gpu

# (P, nVoxel, sVoxel, center, scale_mult)
CASES = [
    (5000, (64, 64, 64), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0), 1.0),          # config A volume
    (20000, (32, 32, 32), (0.25, 0.25, 0.25), (0.1, -0.05, 0.2), 1.0),    # the training-time TV sub-volume query
    (3000, (40, 28, 52), (2.0, 1.4, 2.6), (0.05, 0.0, -0.1), 1.5),        # ragged, anisotropic grid
    (30000, (128, 128, 128), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0), 1.0),
]
IDS = ["A_64cube", "tv_32cube", "ragged_40x28x52", "128cube"]


def _cloud(case):
    P, n, s, ctr, sm = case
    return S.make_cloud(P, seed=P % 89, scale_mult=sm)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_indices_bit_exact(case, oracle, gpu):
    P, n, s, ctr, sm = case
    c = _cloud(case)
    o = Hh.oracle_voxel(oracle, c, n, s, ctr, render=False)
    h = Hh.hip_voxel(c, n, s, ctr, gpu, debug=True)   # debug: introspection-only state (cov3D, inverse permutation) is written
    assert h["num_rendered"] == o["num_rendered"] > 0
    for k in ("radii_x", "radii_y", "radii_z"):
        assert np.array_equal(h[k], o[k]), "%s differs (%d mismatches)" % (k, int((h[k] != o[k]).sum()))
    Hh.check_binning(h, o)
    hp = Hh.hip_voxel(c, n, s, ctr, gpu)
    assert np.array_equal(hp["point_list"], h["point_list"]) and np.array_equal(hp["ranges"], h["ranges"])
    assert np.array_equal(h["cov3D"].view(np.uint32), o["cov3D"].view(np.uint32))
    vis = o["tiles_touched"] > 0
    assert np.array_equal(h["means3D_norm"][vis].view(np.uint32), o["means3D_norm"][vis].view(np.uint32))
    np.testing.assert_allclose(h["conic"][vis], o["conic_opacity"][vis, :6], rtol=3e-7, atol=0)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_volume_within_1e4(case, oracle, gpu):
    P, n, s, ctr, sm = case
    c = _cloud(case)
    o = Hh.oracle_voxel(oracle, c, n, s, ctr)
    h = Hh.hip_voxel(c, n, s, ctr, gpu)
    # pure 1e-4 relative; excesses only on voxels holding a pair ON the 1e-6 cut-off (VOX/forward.cu:293), counted
    st = Hh.parity_volume(oracle, o, h["vol"], "voxel P=%d %s" % (P, "x".join(map(str, n))))
    assert st["n_flip_candidates"] < 0.01 * st["n"]
    assert o["vol"].max() > 0.01


@pytest.mark.parametrize("case", CASES[:3], ids=IDS[:3])
def test_n_contrib_debug_mode(case, oracle, gpu):
    P, n, s, ctr, sm = case
    c = _cloud(case)
    o = Hh.oracle_voxel(oracle, c, n, s, ctr)
    h = Hh.hip_voxel(c, n, s, ctr, gpu, debug=True)
    _budget, nb = oracle.voxel_forward_audit(o)
    mism = h["n_contrib"] != o["n_contrib"]
    assert not (mism & (nb.reshape(-1) == 0)).any(), "unattributed n_contrib mismatches"


@pytest.mark.parametrize("case", CASES[:3], ids=IDS[:3])
def test_backward_vs_oracle(case, oracle, gpu):
    P, n, s, ctr, sm = case
    c = _cloud(case)
    o = Hh.oracle_voxel(oracle, c, n, s, ctr)
    h = Hh.hip_voxel(c, n, s, ctr, gpu)
    g = torch.Generator().manual_seed(1)
    dL = ((torch.rand(*n, generator=g) * 2 - 1) / float(np.prod(n))).numpy()
    gh = Hh.hip_voxel_backward(h, c, n, s, ctr, dL, gpu)
    st = Hh.parity_voxel_grads(oracle, o, gh, c, dL, "voxel P=%d %s" % (P, "x".join(map(str, n))))
    for k in ("dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations"):
        assert st[k]["max_err_over_scale_unflagged"] <= 2e-4, (k, st[k])


def test_empty_and_outside(oracle, gpu):
    from r2_gaussian_amd import _C
    e = torch.empty(0)
    z = lambda *s: torch.zeros(s, device=gpu)
    R, vol, rx, ry, rz, g, b, i = _C.voxelize_gaussians(z(0, 3), z(0, 1), z(0, 3), z(0, 4), 1.0, e, 16, 16, 16, 2.0,
                                                        2.0, 2.0, 0.0, 0.0, 0.0, False, False)
    assert R == 0 and vol.shape == (16, 16, 16) and not vol.any()
    c = S.make_cloud(400, seed=2)
    n, s, ctr = (16, 16, 16), (0.1, 0.1, 0.1), (30.0, 30.0, 30.0)   # volume far from every Gaussian
    o = Hh.oracle_voxel(oracle, c, n, s, ctr)
    h = Hh.hip_voxel(c, n, s, ctr, gpu)
    assert o["num_rendered"] == 0 and h["num_rendered"] == 0 and not h["vol"].any()
    assert not h["radii_x"].any()


def test_sort_key_negative_depth_order(oracle, gpu):
    """The voxelizer's low sort word is raw world-z bits: negative z sorts AFTER positive (quirk Q10)."""
    case = CASES[0]
    P, n, s, ctr, sm = case
    c = _cloud(case)
    h = Hh.hip_voxel(c, n, s, ctr, gpu)
    keys = h["keys"]
    assert (np.diff(keys.astype(np.uint64)) >= 0).all()
    assert (c.xyz[:, 2] < 0).any()


def test_x_slab_sharding_reassembles_the_full_volume(gpu):
    """dist.slab_settings: the full-volume query sharded into x-slabs of whole tiles (one per rank, no exchange).  Every
    slab is an ordinary voxelizer call on a sub-volume; stacked along x they must give the full-volume result (up to the
    float rounding of the shifted voxel coordinates, which can flip a 1e-6 cut-off or a tile of a Gaussian's 3-sigma cube
    for a handful of voxels)."""
    from r2_gaussian_amd import GaussianVoxelizationSettings, GaussianVoxelizer, dist as D
    c = S.make_cloud(20000, seed=4)
    s = GaussianVoxelizationSettings(1.0, 64, 48, 40, 2.0, 1.5, 1.25, 0.0, 0.0, 0.0, False, False)
    args = dict(means3D=c.xyz.to(gpu), opacities=c.density.to(gpu), scales=c.scales.to(gpu), rotations=c.rotations.to(gpu))
    full, _ = GaussianVoxelizer(s)(**args)
    for world in (2, 3, 8):
        parts = []
        for r in range(world):
            sub, (x0, x1) = D.slab_settings(s, r, world)
            if sub is None:
                continue
            vol, _ = GaussianVoxelizer(sub)(**args)
            assert vol.shape == (x1 - x0, 48, 40)
            parts.append(vol)
        got = torch.cat(parts, 0)
        assert got.shape == full.shape
        err = (got - full).abs()
        tol = 1e-4 * full.abs() + 2e-6
        assert float((err > tol).float().mean()) < 1e-4, float((err > tol).float().mean())
        assert float(full.max()) > 0.01


# ---- the small-grid path (csrc/voxel_small.hip: <= 64 tiles, the training loop's TV patch) -----------------------------------
def _small_path_taken(h):
    return int(h["host_words"][2]) == 0x5A11


def test_small_grid_path_is_taken_and_matches_the_general_path(oracle, gpu, monkeypatch):
    """300 k-style sparse patch: the survivor path runs (marker in the state), its point_list / ranges / volume equal the
    general pipeline's (debug mode forces that one) bit for bit, and the oracle's."""
    c = S.make_cloud(60000, seed=13)
    n, s, ctr = (32, 32, 32), (0.25, 0.25, 0.25), (-0.2, 0.1, 0.0)
    fast = Hh.hip_voxel(c, n, s, ctr, gpu)
    general = Hh.hip_voxel(c, n, s, ctr, gpu, debug=True)
    assert _small_path_taken(fast) and not _small_path_taken(general)
    assert fast["num_rendered"] == general["num_rendered"] > 0
    for k in ("radii_x", "radii_y", "radii_z", "tiles_touched", "point_list", "ranges"):
        assert np.array_equal(fast[k], general[k]), k
    # (the debug forward renders voxel-parallel, the production one lane-per-entry: same lists, another association of the sums)
    np.testing.assert_allclose(fast["vol"], general["vol"], rtol=2e-5, atol=1e-9)
    o = Hh.oracle_voxel(oracle, c, n, s, ctr, render=False)
    Hh.check_binning(fast, o)
    # ragged small grid with empty tiles: 20 x 12 x 28 voxels = 3 x 2 x 4 tiles, most of them outside the cloud
    n2, s2, ctr2 = (20, 12, 28), (0.2, 0.12, 0.28), (1.08, 1.02, 0.9)
    f2 = Hh.hip_voxel(c, n2, s2, ctr2, gpu)
    o2 = Hh.oracle_voxel(oracle, c, n2, s2, ctr2, render=False)
    assert _small_path_taken(f2)
    Hh.check_binning(f2, o2)
    assert f2["num_rendered"] > 0


def test_small_grid_equal_depth_keys_keep_id_order(oracle, gpu):
    """All Gaussians on one z plane: every sort key is equal, the reference's stable sort leaves ascending ids; the in-LDS bucket
    sort then has ONE long bucket and must still rank by (key, id)."""
    c = S.make_cloud(20000, seed=5, scale_mult=0.3)
    xyz = torch.cat([c.xyz[:, :2] * 0.1, torch.full((20000, 1), 0.0123)], 1)
    xyz[3000:, 0] += 50.0                       # only the first 3000 reach the patch (the state's temp is sized by P)
    c = S.Cloud(xyz.contiguous(), c.scales, c.rotations, c.density)
    n, s, ctr = (32, 32, 32), (0.3, 0.3, 0.3), (0.0, 0.0, 0.0)
    o = Hh.oracle_voxel(oracle, c, n, s, ctr, render=False)
    h = Hh.hip_voxel(c, n, s, ctr, gpu)
    assert _small_path_taken(h) and h["num_rendered"] > 5000
    Hh.check_binning(h, o)
    for a, b in h["ranges"]:
        assert (np.diff(h["point_list"][a:b].astype(np.int64)) > 0).all()   # equal keys: ids ascend inside every tile


def test_small_grid_falls_back_when_the_patch_holds_too_many(oracle, gpu):
    """More survivors than the LDS sort holds (8192): the call silently takes the general pipeline; same results."""
    c = S.make_cloud(40000, seed=3)
    n, s, ctr = (32, 32, 32), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0)      # the patch is the whole volume: everybody survives
    o = Hh.oracle_voxel(oracle, c, n, s, ctr)
    h = Hh.hip_voxel(c, n, s, ctr, gpu)
    assert not _small_path_taken(h) and h["num_rendered"] == o["num_rendered"]
    Hh.check_binning(h, o)
    Hh.parity_volume(oracle, o, h["vol"], "small-grid fallback 40k/32^3")
    g = torch.Generator().manual_seed(2)
    dL = ((torch.rand(*n, generator=g) * 2 - 1) / float(np.prod(n))).numpy()
    gh = Hh.hip_voxel_backward(h, c, n, s, ctr, dL, gpu)
    st = Hh.parity_voxel_grads(oracle, o, gh, c, dL, "small-grid fallback 40k/32^3")
    for k in ("dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations"):
        assert st[k]["max_err_over_scale_unflagged"] <= 2e-4, (k, st[k])


def test_small_grid_from_two_threads(gpu):
    """The path keeps a per-host-thread device counter: two threads hammering different patches on their own streams."""
    import threading
    from r2_gaussian_amd import GaussianVoxelizationSettings, GaussianVoxelizer
    c = S.make_cloud(50000, seed=8)
    args = dict(means3D=c.xyz.to(gpu), opacities=c.density.to(gpu), scales=c.scales.to(gpu), rotations=c.rotations.to(gpu))
    centres = [(-0.3 + 0.1 * i, 0.05 * i - 0.1, 0.02 * i) for i in range(6)]
    mk = lambda ctr: GaussianVoxelizer(GaussianVoxelizationSettings(1.0, 32, 32, 32, 0.25, 0.25, 0.25, ctr[0], ctr[1], ctr[2], False, False))
    ref = [mk(ctr)(**args)[0].clone() for ctr in centres]
    torch.cuda.synchronize()
    bad = []

    def worker(t):
        st = torch.cuda.Stream(device=gpu)
        with torch.cuda.stream(st):
            for rep in range(30):
                i = (rep * 2 + t) % len(centres)
                vol, _ = mk(centres[i])(**args)
                st.synchronize()
                if not torch.equal(vol, ref[i]):
                    bad.append((t, rep, i))
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not bad, bad[:5]


def test_backward_after_a_hinted_overflow(oracle, gpu):
    """when the hinted depth order overflows its buckets the forward falls back to a radix sort of ALL P depth
    keys -- culled Gaussians interleaved, their key is bits(z), not a sentinel -- while the dual scan had already left its visible
    count in the state; the geometry backward then walked only that prefix of `order` and visible Gaussians sorted behind it got
    no gradient rows (uninitialised memory on large grids).  Force exactly that: a hint armed by a wide cloud, then the same P
    with z squeezed into a handful of buckets and half of the cloud outside the volume; grid > 64 tiles (not the small path)."""
    from r2_gaussian_amd import _lib
    L = _lib.lib()
    P = 20000
    c = S.make_cloud(P, seed=6)
    n, s, ctr = (40, 40, 40), (1.0, 1.0, 1.0), (0.5, 0.0, 0.0)     # x < 0 is outside: about half of the cloud is culled
    try:
        L.r2_voxel_sticks_control(0)                               # the general binning chain is the one with a depth order
        L.r2_depth_hint_control(2)
        Hh.hip_voxel(c, n, s, ctr, gpu)                            # un-hinted, arms the hint for this P
        xyz = c.xyz.clone()
        xyz[:, 2] = xyz[:, 2] * 1e-5 + 0.01
        c3 = S.Cloud(xyz, c.scales, c.rotations, c.density)
        o = Hh.oracle_voxel(oracle, c3, n, s, ctr)
        nvis = int((o["tiles_touched"] > 0).sum())
        assert 0.2 * P < nvis < 0.8 * P
        h = Hh.hip_voxel(c3, n, s, ctr, gpu)
        assert int(h["host_words"][1]) == 1, "the squeezed cloud was meant to overflow the hinted buckets"
        assert int(h["host_words"][7]) == 0, "after the fallback `order` holds all P ids: the visible-prefix count must be cleared"
        Hh.check_binning(h, o)
        g = torch.Generator().manual_seed(3)
        dL = ((torch.rand(*n, generator=g) * 2 - 1) / float(np.prod(n))).numpy()
        for k in range(2):   # twice: the second backward runs on recycled (dirty) gradient buffers
            gh = Hh.hip_voxel_backward(h, c3, n, s, ctr, dL, gpu)
            sg = Hh.parity_voxel_grads(oracle, o, gh, c3, dL, "voxel backward after a hinted overflow (%d)" % k)
            for kk in ("dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations"):
                assert sg[kk]["max_err_over_scale_unflagged"] <= 2e-4, (kk, sg[kk])
            # every visible Gaussian got a row
            assert (np.abs(gh["dL_dopacity"].reshape(-1)[o["tiles_touched"] > 0]) > 0).mean() > 0.9
    finally:
        L.r2_voxel_sticks_control(1)
        L.r2_depth_hint_control(1)
        L.r2_depth_hint_control(2)


@pytest.mark.parametrize("n,s,ctr,P,mult", [((20, 13, 27), (0.5, 0.4, 0.6), (0.1, 0.0, -0.1), 8000, 1.0),
                                            ((32, 32, 32), (0.25, 0.25, 0.25), (-0.2, 0.1, 0.0), 60000, 3.0),
                                            ((8, 8, 8), (0.06, 0.06, 0.06), (0.0, 0.0, 0.0), 30000, 1.0)],
                         ids=["ragged-20x13x27", "big-gaussians-32cube", "one-tile"])
def test_patch_backward_ragged_big_and_single_tile(n, s, ctr, P, mult, oracle, gpu):
    """Backward on patches (<= 64 tiles; since round 4 the gradient zero-fill rides along with the render backward's launch):
    ragged grids (rows that are not whole float4s, tiles that stick out of the volume), Gaussians spanning the whole patch, a
    single tile; twice, on recycled gradient buffers: every culled row zero, results bit-reproducible."""
    c = S.make_cloud(P, seed=17, scale_mult=mult)
    o = Hh.oracle_voxel(oracle, c, n, s, ctr)
    h = Hh.hip_voxel(c, n, s, ctr, gpu)
    assert h["num_rendered"] == o["num_rendered"] > 0
    g = torch.Generator().manual_seed(5)
    dL = ((torch.rand(*n, generator=g) * 2 - 1) / float(np.prod(n))).numpy()
    first = None
    for rep in range(2):
        gh = Hh.hip_voxel_backward(h, c, n, s, ctr, dL, gpu)
        sg = Hh.parity_voxel_grads(oracle, o, gh, c, dL, "patch backward %s (%d)" % ("x".join(map(str, n)), rep))
        for k in ("dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations"):
            assert sg[k]["max_err_over_scale_unflagged"] <= 2e-4, (k, sg[k])
        assert sg["after_flips"]["max_err_over_tol_after_flips"] <= 0.3, sg["after_flips"]
        culled = o["tiles_touched"] == 0
        for k in gh:
            assert not gh[k].reshape(P, -1)[culled].any(), k      # every culled row is a zero row
            if first is not None:
                assert np.array_equal(gh[k], first[k]), k          # bit-reproducible
        first = gh
