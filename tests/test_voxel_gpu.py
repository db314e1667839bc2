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


def _oracle_lists_of_slab(o, grid, t0, t1):
    """The oracle's (full-grid) tile lists restricted to the tile layers [t0, t1) along x, tiles renumbered from t0: what the
    x-slab call must produce -- point_list, ranges, the sorted (tile | z bits) keys."""
    gx, gy, gz = grid
    sx = t1 - t0
    ranges = np.zeros((sx * gy * gz, 2), np.uint32)
    pl, keys, pos = [], [], 0
    dk = o["depths"].view(np.uint32)
    for z in range(gz):
        for y in range(gy):
            for x in range(t0, t1):
                a, b = (int(v) for v in o["ranges"][(z * gy + y) * gx + x])
                if b > a:
                    t = (z * gy + y) * sx + (x - t0)
                    ids = o["point_list"][a:b]
                    ranges[t] = (pos, pos + b - a)
                    pl.append(ids)
                    keys.append((np.uint64(t) << np.uint64(32)) | dk[ids].astype(np.uint64))
                    pos += b - a
    cat = lambda v, dt: np.concatenate(v) if v else np.zeros((0,), dt)   # noqa: E731
    return cat(pl, np.uint32), ranges, cat(keys, np.uint64), pos


def _check_slabs(gpu, oracle, c, n, s, ctr, worlds, full_vol=None):
    from r2_gaussian_amd import GaussianVoxelizationSettings, dist as D
    o = Hh.oracle_voxel(oracle, c, n, s, ctr, render=False)
    grid = tuple((k + 7) // 8 for k in n)
    if full_vol is None:
        full_vol = Hh.hip_voxel(c, n, s, ctr, gpu)["vol"]
    st = GaussianVoxelizationSettings(1.0, n[0], n[1], n[2], s[0], s[1], s[2], ctr[0], ctr[1], ctr[2], False, False)
    full_radii = np.stack([o["radii_x"], o["radii_y"], o["radii_z"]])
    for world in worlds:
        parts, total = [], 0
        for r in range(world):
            sub, (x0, x1) = D.slab_settings(st, r, world)
            if sub is None:
                continue
            h = Hh.hip_voxel(c, n, s, ctr, gpu, slab=(sub.tile_x0, sub.tile_x1))
            pl, ranges, keys, R = _oracle_lists_of_slab(o, grid, sub.tile_x0, sub.tile_x1)
            assert h["num_rendered"] == R, (world, r, h["num_rendered"], R)
            assert np.array_equal(h["point_list"], pl), "slab point_list != the full lists restricted to the slab"
            assert np.array_equal(h["ranges"], ranges), "slab ranges differ"
            assert np.array_equal(h["keys"], keys), "slab (tile | z bits) keys differ"
            # radii: the full call's for the Gaussians with a tile in the slab, 0 for the others
            got_r = np.stack([h["radii_x"], h["radii_y"], h["radii_z"]])
            inside = h["tiles_touched"] > 0
            assert np.array_equal(got_r[:, inside], full_radii[:, inside]) and not got_r[:, ~inside].any()
            parts.append(h["vol"])
            total += R
        assert total == o["num_rendered"], "the slabs' instances do not add up to the full call's"
        got = np.concatenate(parts, 0)
        assert got.shape == full_vol.shape
        assert np.array_equal(got.view(np.uint32), full_vol.view(np.uint32)), \
            "world %d: concatenated slabs are not bit-identical to the unsharded volume (%d voxels differ)" % (
                world, int((got.view(np.uint32) != full_vol.view(np.uint32)).sum()))
    return full_vol


def test_x_slab_sharding_is_bit_identical_to_the_unsharded_query(oracle, gpu):
    """The sharded full-volume query (test.py:105-112, VOX/forward.cu:58-178, VOX/voxelizer_impl.cu:54-101): every rank
    runs the FULL grid's arithmetic and renders its tile layers only (r2_voxel_forward_slab).  torch.equal(cat(slabs), full) for
    world in {2, 3, 8}, every slab's point_list / ranges / keys = the oracle's lists restricted to the slab -- no tolerance.
    (Rounds 1-5 re-centred a sub-volume per rank and had to tolerate 1e-4 of the voxels outside 1e-4 relative.)"""
    c = S.make_cloud(20000, seed=4)
    full = _check_slabs(gpu, oracle, c, (64, 48, 40), (2.0, 1.5, 1.25), (0.0, 0.0, 0.0), (2, 3, 8))
    assert float(full.max()) > 0.01
    # an off-centre volume with a ragged last layer (nx = 52: 7 layers, the last one 4 voxels thick) and more ranks than layers
    _check_slabs(gpu, oracle, c, (52, 40, 24), (1.7, 1.3, 0.8), (0.11, -0.07, 0.05), (2, 5, 9))


def test_x_slab_sharding_is_bit_identical_at_256_cubed(oracle, gpu):
    """... and at the reported size: 300k Gaussians, the 256^3 query, world 8 (slabs of 4096 tiles on the stick-first chain) and 3."""
    c = S.make_cloud(300000, seed=0)
    _check_slabs(gpu, oracle, c, (256, 256, 256), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0), (8, 3))


def test_x_slab_backward_adds_up_to_the_full_gradient(gpu):
    """A slab call is differentiable like any other: with dL/dvol cut the same way, the slabs' parameter gradients add up to the
    full call's (float association differs: every slab reduces its own instances)."""
    from r2_gaussian_amd import GaussianVoxelizationSettings, GaussianVoxelizer, dist as D
    c = S.make_cloud(6000, seed=9)
    st = GaussianVoxelizationSettings(1.0, 48, 40, 32, 1.8, 1.5, 1.2, 0.0, 0.0, 0.0, False, False)
    g = torch.Generator().manual_seed(1)
    dL = torch.rand((48, 40, 32), generator=g).to(gpu)

    def run(settings, dl):
        leaves = [t.to(gpu).clone().requires_grad_(True) for t in (c.xyz, c.density, c.scales, c.rotations)]
        vol, _ = GaussianVoxelizer(settings)(means3D=leaves[0], opacities=leaves[1], scales=leaves[2], rotations=leaves[3])
        vol.backward(dl)
        return vol.detach(), [t.grad.double() for t in leaves]
    full, gfull = run(st, dL)
    acc = [torch.zeros_like(t) for t in gfull]
    vols = []
    for r in range(3):
        sub, (x0, x1) = D.slab_settings(st, r, 3)
        v, gs = run(sub, dL[x0:x1].contiguous())
        vols.append(v)
        acc = [a + b for a, b in zip(acc, gs)]
    assert torch.equal(torch.cat(vols, 0), full)
    for a, b, name in zip(acc, gfull, ("xyz", "density", "scales", "rotations")):
        scale = float(b.abs().max())
        assert scale > 0 and float((a - b).abs().max()) <= 2e-5 * scale, (name, float((a - b).abs().max()), scale)


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
