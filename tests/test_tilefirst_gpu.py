"""GPU: the tile-first binning chain of the rasterizer forward (csrc/raster_tilefirst.hip, round 4).

A forward takes it when the calling thread can predict the call's instance count from its recent calls with the same P and
detector (the first call of a size runs the general chain and leaves the prediction behind).  Contract: num_rendered, radii,
tiles_touched, point_list, ranges bit-identical to the oracle (= the reference's (tile | depth) order) and to the general chain;
the image bit-identical to the general chain's (same lists, same work items, same summation order); every visible Gaussian owns
a run of backward scratch rows, the runs a partition of [0, R); the backward unchanged.

Cases: ordinary scenes incl. 4096 tiles; tile lists beyond one sort part (several workgroups per tile, split by depth range);
thousands of equal depths in one tile (the rank-by-counting fallback); a prediction that falls short (second pass with exact
sizes); a scene that turns out to hold thin Gaussians after the render was enqueued without that variant; nothing visible; two
host threads.
"""
import threading

import numpy as np
import pytest
import torch

from r2_gaussian_amd import scene as S
from tests import helpers as Hh

pytestmark = pytest.mark.gpu


@pytest.fixture()
def L():
    from r2_gaussian_amd import _lib
    lib = _lib.lib()
    lib.r2_tile_first_control(1)
    lib.r2_tile_first_control(2)
    yield lib
    lib.r2_tile_first_control(1)
    lib.r2_tile_first_control(2)


def _both(L, c, v, gpu, **kw):
    """-> (general chain's result, tile-first result) for the same call."""
    L.r2_tile_first_control(0)
    g = Hh.hip_raster(c, v, gpu, **kw)
    assert not Hh.took_tile_first(g)
    L.r2_tile_first_control(1)
    L.r2_tile_first_control(2)
    Hh.hip_raster(c, v, gpu, **kw)            # leaves the prediction
    t = Hh.hip_raster(c, v, gpu, **kw)
    assert Hh.took_tile_first(t), "the second call of a size did not take the tile-first chain"
    return g, t


def _same(g, t):
    assert t["num_rendered"] == g["num_rendered"]
    for k in ("radii", "tiles_touched", "point_list", "ranges", "depth_key", "color"):
        assert np.array_equal(g[k], t[k]), k
    vis = t["tiles_touched"] > 0
    for k in ("cov3D", "rec"):                    # (rows of culled Gaussians are never written)
        assert np.array_equal(g[k][vis], t[k][vis]), k
    assert int(t["host_words"][7]) == int((t["tiles_touched"] > 0).sum())


@pytest.mark.parametrize("P,det,seed,mult", [(7, (48, 48), 3, 1.0), (3000, (64, 100), 8, 1.0), (30000, (160, 144), 21, 1.0),
                                             (120000, (512, 512), 5, 1.0), (60000, (512, 512), 6, 3.0), (400000, (1024, 1024), 2, 1.0)],
                         ids=["tiny", "3k", "30k", "120k-512", "60k-big-gaussians", "400k-1024"])
def test_identical_to_the_general_chain_and_the_oracle(P, det, seed, mult, L, oracle, gpu):
    c = S.make_cloud(P, seed=seed, scale_mult=mult)
    v = S.make_views(8, det)[3]
    g, t = _both(L, c, v, gpu)
    _same(g, t)
    o = Hh.oracle_raster(oracle, c, v, render=False)
    Hh.check_binning(t, o)
    if det == (1024, 1024):
        assert o["ranges"].shape[0] == 4096


def test_long_lists_are_split_by_depth_range(L, oracle, gpu):
    """Tile lists of > 4096 entries are sorted by several workgroups, each a range of the list's depth histogram."""
    c = S.make_cloud(300000, seed=0)
    v = S.make_views(50, (512, 512))[0]
    g, t = _both(L, c, v, gpu)
    _same(g, t)
    lens = t["ranges"][:, 1].astype(np.int64) - t["ranges"][:, 0]
    assert lens.max() > 2 * 4096, "the headline scene was meant to hold lists of three parts"
    o = Hh.oracle_raster(oracle, c, v, render=False)
    Hh.check_binning(t, o)
    # ... and a list that needs many parts with a lopsided depth distribution: half of the cloud squeezed into a thin slab
    xyz = c.xyz.clone()
    xyz[::2, 0] = xyz[::2, 0] * 0.02 + 0.3
    c2 = S.Cloud(xyz, c.scales * 2.0, c.rotations, c.density)
    v0 = S.make_view(0.0, (256, 256))          # looks along -x: depth = 5 - x
    g, t = _both(L, c2, v0, gpu)
    _same(g, t)
    assert (t["ranges"][:, 1].astype(np.int64) - t["ranges"][:, 0]).max() > 20000
    Hh.check_binning(t, Hh.oracle_raster(oracle, c2, v0, render=False))


def test_thousands_of_equal_depths_in_one_tile(L, oracle, gpu):
    """All Gaussians at the same view depth: the depth histogram cannot split the lists, ids decide (the reference's tie rule);
    parts too large for the LDS sort are ranked by counting."""
    P = 20000
    c = S.make_cloud(P, seed=11)
    xyz = c.xyz.clone()
    xyz[:, 0] = 0.25                             # view 0 looks along -x: one depth for all
    xyz[:, 1:] *= 0.2                            # ... on a few tiles
    c2 = S.Cloud(xyz, c.scales, c.rotations, c.density)
    v = S.make_view(0.0, (128, 128))
    g, t = _both(L, c2, v, gpu)
    _same(g, t)
    o = Hh.oracle_raster(oracle, c2, v, render=False)
    assert len(np.unique(o["depths"][o["tiles_touched"] > 0])) == 1
    assert (o["ranges"][:, 1].astype(np.int64) - o["ranges"][:, 0]).max() > 6144
    Hh.check_binning(t, o)


def test_a_prediction_that_falls_short_costs_a_second_pass_only(L, oracle, gpu):
    P = 50000
    small = S.make_cloud(P, seed=4, scale_mult=0.5)
    big = S.make_cloud(P, seed=4, scale_mult=3.0)
    v = S.make_views(8, (256, 256))[1]
    Hh.hip_raster(small, v, gpu)                 # the prediction: a few instances per Gaussian
    t = Hh.hip_raster(big, v, gpu)               # ... and then 30x as many
    assert Hh.took_tile_first(t)
    o = Hh.oracle_raster(oracle, big, v)
    assert o["num_rendered"] > 4 * Hh.oracle_raster(oracle, small, v, render=False)["num_rendered"]
    Hh.check_binning(t, o)
    Hh.parity_image(oracle, o, t["color"], "tile-first after a short prediction")
    dL = S.make_pixel_grad(256, 256).numpy()
    gh = Hh.hip_raster_backward(t, big, v, dL, gpu)
    Hh.parity_raster_grads(oracle, o, gh, big, v, dL, "tile-first after a short prediction")
    # the other way round (far fewer instances than predicted) needs nothing special
    t2 = Hh.hip_raster(small, v, gpu)
    assert Hh.took_tile_first(t2)
    Hh.check_binning(t2, Hh.oracle_raster(oracle, small, v, render=False))


def test_thin_gaussians_need_no_second_render(L, oracle, gpu):
    """One forward variant since round 6: Gaussians too thin for the recurrences (item_tier) take the kernels' exact path, whatever
    the thread saw before (rounds 4-5 chose between two variants from the previous call's flag and rendered AGAIN after a wrong guess).
    Image and every gradient are, bit for bit, what the general chain produces."""
    import ctypes as C
    P = 20000
    plain = S.make_cloud(P, seed=9)
    v = S.make_views(8, (192, 192))[5]
    h0 = Hh.hip_raster(plain, v, gpu)
    n0 = int(h0["host_words"][2])                # Gaussians on the exact path (item_tier), as the preprocess counts them: few here
    assert n0 < 500, n0
    sc = plain.scales.clone()
    sc[:8000] *= 0.12                            # sub-pixel Gaussians: conditional sigma ~0.4 px
    thin = S.Cloud(plain.xyz, sc, plain.rotations, plain.density)
    st = (C.c_longlong * 5)()
    L.r2_tile_first_stats(st, 1)
    t = Hh.hip_raster(thin, v, gpu)
    nthin = int(t["host_words"][2])
    assert Hh.took_tile_first(t) and nthin > n0 + 1000, "the scene was meant to hold thin Gaussians: %d (plain cloud: %d)" % (nthin, n0)
    L.r2_tile_first_stats(st, 0)
    assert list(st)[2:4] == [0, 0], "no second pass, no repeated render"
    o = Hh.oracle_raster(oracle, thin, v)
    Hh.check_binning(t, o)
    Hh.parity_image(oracle, o, t["color"], "tile-first, thin Gaussians")
    dL = S.make_pixel_grad(192, 192).numpy()
    gh = Hh.hip_raster_backward(t, thin, v, dL, gpu)
    L.r2_tile_first_control(0)
    g = Hh.hip_raster(thin, v, gpu)
    assert not Hh.took_tile_first(g) and np.array_equal(g["color"], t["color"])
    gg = Hh.hip_raster_backward(g, thin, v, dL, gpu)
    for k in gh:
        assert np.array_equal(gh[k], gg[k]), k


def test_forward_backward_parity_and_nothing_visible(L, oracle, gpu):
    c = S.make_cloud(40000, seed=13)
    v = S.make_views(8, (256, 200))[6]
    Hh.hip_raster(c, v, gpu)
    for rep in range(2):                          # the second one runs on the self-reset counters of the first
        t = Hh.hip_raster(c, v, gpu)
        assert Hh.took_tile_first(t)
        o = Hh.oracle_raster(oracle, c, v)
        Hh.check_binning(t, o)
        Hh.parity_image(oracle, o, t["color"], "tile-first %d" % rep)
        dL = S.make_pixel_grad(256, 200).numpy()
        gh = Hh.hip_raster_backward(t, c, v, dL, gpu)
        sg = Hh.parity_raster_grads(oracle, o, gh, c, v, dL, "tile-first %d" % rep)
        assert sg["after_flips"]["max_err_over_tol_after_flips"] <= 0.3
    # every Gaussian behind the source: nothing visible, every pixel zero
    xyz = c.xyz.clone()
    xyz[:, 0] += 20.0
    gone = S.Cloud(xyz, c.scales, c.rotations, c.density)
    v0 = S.make_view(0.0, (256, 200))
    Hh.hip_raster(gone, v0, gpu)
    t = Hh.hip_raster(gone, v0, gpu)
    assert t["num_rendered"] == 0 and not t["color"].any() and not t["radii"].any()
    t = Hh.hip_raster(c, v0, gpu)                 # and the counters are clean afterwards
    Hh.check_binning(t, Hh.oracle_raster(oracle, c, v0, render=False))


def test_two_host_threads(L, oracle, gpu):
    """Predictions and counters are per host thread (and stream): two threads in the forward at once."""
    clouds = [S.make_cloud(25000, seed=31), S.make_cloud(26000, seed=32)]
    v = S.make_views(8, (160, 160))[2]
    refs = [Hh.oracle_raster(oracle, c, v, render=False) for c in clouds]
    errs, took = [], []

    def worker(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=gpu)):
                for _ in range(6):
                    h = Hh.hip_raster(clouds[i], v, gpu)
                    Hh.check_binning(h, refs[i])
                took.append(Hh.took_tile_first(h))
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errs, errs
    assert took == [True, True]


def test_a_new_gaussian_count_is_seeded_from_the_previous_one(L, oracle, gpu):
    """Round 5.  Every densification changes P (train.py:155-168); round 4 sent the first call of every new P
    through the general chain.  Now its prediction is seeded from the thread's last call on the same detector (instances per
    Gaussian x the new count), so the call after a densification takes the tile-first chain as well -- and a seed that falls short
    (the new Gaussians are much larger) costs a second pass, never a wrong result."""
    import ctypes
    v = S.make_views(8, (256, 256))[2]
    full = S.make_cloud(60000, seed=17)
    stats = (ctypes.c_longlong * 5)()

    def prefix(n, mult=1.0):
        return S.Cloud(full.xyz[:n].contiguous(), (full.scales[:n] * mult).contiguous(), full.rotations[:n].contiguous(),
                       full.density[:n].contiguous())
    Hh.hip_raster(prefix(20000), v, gpu)                       # first call on this detector: general chain, leaves a prediction
    L.r2_tile_first_stats(stats, 1)
    for n in (22000, 26000, 31000, 40000, 60000):              # five "densifications"
        c = prefix(n)
        t = Hh.hip_raster(c, v, gpu)
        assert Hh.took_tile_first(t), "the first call at P = %d did not take the tile-first chain" % n
        Hh.check_binning(t, Hh.oracle_raster(oracle, c, v, render=False))
    L.r2_tile_first_stats(stats, 0)
    assert stats[0] == 5 and stats[1] == 0 and stats[4] == 5, list(stats)
    # a seed that is far too small: the same count of Gaussians three times the size
    L.r2_tile_first_stats(stats, 1)
    big = prefix(61000, mult=3.0)
    t = Hh.hip_raster(big, v, gpu)
    L.r2_tile_first_stats(stats, 0)
    assert Hh.took_tile_first(t) and stats[2] == 1, list(stats)
    o = Hh.oracle_raster(oracle, big, v)
    Hh.check_binning(t, o)
    Hh.parity_image(oracle, o, t["color"], "tile-first, seeded prediction that fell short")



def test_deferred_count_returns_a_token_and_the_backward_resolves_it(L, gpu):
    """r2_defer_count_control(1) (SURVEY 7 "kill the D2H sync", RAS/rasterizer_impl.cu:279): the forward returns
    without waiting; what it hands back in place of num_rendered is a token that the backward turns into the count.  Image, radii,
    state and every gradient are the waiting mode's, bit for bit; a prediction that falls short makes the BACKWARD fail loudly."""
    import ctypes as C
    from r2_gaussian_amd import _C, _lib
    P = 30000
    c = S.make_cloud(P, seed=12)
    views = S.make_views(8, (256, 256))
    dL = S.make_pixel_grad(256, 256).numpy()
    Hh.hip_raster(c, views[0], gpu)                      # a first call of this size: the general chain leaves the prediction
    ref = Hh.hip_raster(c, views[3], gpu)                # waiting mode, tile-first chain
    assert Hh.took_tile_first(ref) and ref["num_rendered"] < 0x40000000
    gref = Hh.hip_raster_backward(ref, c, views[3], dL, gpu)
    st = (C.c_longlong * 3)()
    L.r2_defer_count_stats(st, 1)
    L.r2_defer_count_control(1)
    try:
        t = Hh.hip_raster(c, views[3], gpu, count_of=ref["num_rendered"])
        assert t["num_rendered"] >= 0x40000000, "the forward did not return a token"
        assert np.array_equal(t["color"].view(np.uint32), ref["color"].view(np.uint32)) and np.array_equal(t["radii"], ref["radii"])
        assert np.array_equal(t["point_list"], ref["point_list"]) and np.array_equal(t["ranges"], ref["ranges"])
        assert int(t["host_words"][0]) == ref["num_rendered"]
        g = Hh.hip_raster_backward(t, c, views[3], dL, gpu)   # num_rendered = the token
        for k in gref:
            assert np.array_equal(g[k].view(np.uint32), gref[k].view(np.uint32)), k
        L.r2_defer_count_stats(st, 0)
        assert list(st) == [1, 0, 0]
        # several forwards in flight before their backwards (two views per optimiser step), resolved in any order
        ts = [Hh.hip_raster(c, views[i], gpu, count_of=None) for i in (1, 2)]
        gs = [Hh.hip_raster_backward(ts[i], c, views[(1, 2)[i]], dL, gpu) for i in (1, 0)]
        L.r2_defer_count_control(0)
        for i, vi in ((1, 2), (0, 1)):
            w = Hh.hip_raster(c, views[vi], gpu)
            gw = Hh.hip_raster_backward(w, c, views[vi], dL, gpu)
            got = gs[0] if i == 1 else gs[1]
            assert np.array_equal(got["dL_dmeans3D"].view(np.uint32), gw["dL_dmeans3D"].view(np.uint32))
        # a prediction that falls short: 30x the instances.  The deferred forward cannot repair it; its backward says so
        big = S.make_cloud(P, seed=12, scale_mult=3.0)
        L.r2_defer_count_control(1)
        e = torch.empty(0)
        v3 = views[3]
        args = (big.xyz.to(gpu), big.density.to(gpu), big.scales.to(gpu), big.rotations.to(gpu), 1.0, e, v3.world_view_transform.to(gpu),
                v3.full_proj_transform.to(gpu), v3.tanfovx, v3.tanfovy, v3.image_height, v3.image_width, v3.camera_center.to(gpu),
                False, v3.mode, False)
        Rb, _color, radii_b, gb, bb, ib = _C.rasterize_gaussians(*args)   # (its state is invalid: not read back)
        assert Rb >= 0x40000000
        with pytest.raises(_lib.R2HipError, match="sized for"):
            _C.rasterize_gaussians_backward(args[0], radii_b, args[2], args[3], 1.0, e, args[6], args[7], args[8], args[9],
                                            torch.as_tensor(dL).to(gpu), args[12], gb, Rb, bb, ib, v3.mode, False)
        torch.cuda.synchronize()
        L.r2_defer_count_control(0)
        # ... and the thread's next call is none the worse for it (the counters the failed call left behind are cleaned)
        again = Hh.hip_raster(c, views[3], gpu)
        assert np.array_equal(again["color"].view(np.uint32), ref["color"].view(np.uint32))
        assert np.array_equal(again["point_list"], ref["point_list"])
    finally:
        L.r2_defer_count_control(0)


def test_gaussians_on_both_sides_of_the_recurrence_tier_match_the_oracle(L, oracle, gpu):
    """Round 6: the render kernels walk a block's rows by a second recurrence; item_tier (csrc/raster_state.hpp) sends the entries the
    two walks are not safe for down the exact path.  A cloud whose projected sigmas straddle that boundary (~0.5 to 2 px: both
    paths well populated, and the recurrence path at the thin end of what it accepts, where its error is largest): image and
    every gradient against the oracle at the usual tolerance, lists bit-exact."""
    c = S.make_cloud(20000, seed=33, scale_mult=0.55)
    v = S.make_views(8, (192, 192))[2]
    Hh.hip_raster(c, v, gpu)
    t = Hh.hip_raster(c, v, gpu)
    assert Hh.took_tile_first(t)
    nvis, nexact = int((t["radii"] > 0).sum()), int(t["host_words"][2])
    assert 0.05 * nvis < nexact < 0.95 * nvis, "the cloud was meant to straddle the tier: %d of %d visible on the exact path" % (nexact, nvis)
    o = Hh.oracle_raster(oracle, c, v)
    Hh.check_binning(t, o)
    Hh.parity_image(oracle, o, t["color"], "both sides of item_tier")
    dL = S.make_pixel_grad(192, 192).numpy()
    gh = Hh.hip_raster_backward(t, c, v, dL, gpu)
    Hh.parity_raster_grads(oracle, o, gh, c, v, dL, "both sides of item_tier")
