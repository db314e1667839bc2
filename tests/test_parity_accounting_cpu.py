"""CPU: the parity accounting itself (oracle/parity.py + the AUDIT section of oracle/r2_oracle.c).

* the REFERENCE's own kernels run on the CPU (oracle/_ref, float atomics in fiber order) must pass the gradient bound
  rtol * sum|terms| + flip budget against the oracle's double sums -- i.e. the bound is not tighter than what the
  reference itself delivers;
* an image evaluated the way the HIP kernels do it (float32 exp2 of a pre-scaled conic, different rounding than expf)
  passes the pure 1e-4 bound everywhere except on attributed cut-off flips, and a fabricated non-borderline error is caught;
* the audit's double sums agree with the acc64 oracle path.
"""
import numpy as np
import pytest

from oracle import parity as Pz
from r2_gaussian_amd import scene as S
from tests import helpers as Hh

LOG2E = np.float32(1.4426950408889634)


def _scene(P=4000, hw=(96, 80), seed=3, sm=1.0, angle=1.1):
    c = S.make_cloud(P, seed=seed, scale_mult=sm)
    v = S.make_view(angle, hw)
    return c, v


def test_audit_sums_match_acc64(oracle):
    c, v = _scene()
    o = Hh.oracle_raster(oracle, c, v)
    dL = S.make_pixel_grad(v.image_height, v.image_width).numpy()
    xyz, rho, sc, q = Hh.cloud_np(c)
    vm, pm = Hh.np_view(v)
    g = oracle.raster_backward(o, xyz, sc, q, 1.0, None, vm, pm, v.tanfovx, v.tanfovy, dL, acc64=True)
    s, a, f = oracle.raster_backward_audit(o, dL)
    np.testing.assert_allclose(g["dL_dmeans2D"][:, :2], s[:, :2].astype(np.float32), rtol=2e-6, atol=1e-12)
    np.testing.assert_allclose(g["dL_dopacity"][:, 0], s[:, 5].astype(np.float32), rtol=2e-6, atol=1e-12)
    np.testing.assert_allclose(g["dL_dconic"].reshape(-1, 4)[:, [0, 1, 3]], s[:, 2:5].astype(np.float32), rtol=2e-6, atol=1e-12)
    assert (a >= np.abs(s) * (1 - 1e-12)).all() and (f >= 0).all()
    # the chain on the audit's sums reproduces the oracle's final gradients
    ch = oracle.raster_geom_chain(o, s.astype(np.float32), xyz, sc, q, 1.0, None, vm, pm, v.tanfovx, v.tanfovy)
    for k in ("dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations"):
        scale = np.abs(g[k]).max()
        assert np.abs(ch[k] - g[k]).max() <= 1e-5 * scale, k


def test_reference_float_atomics_pass_the_gradient_bound(oracle):
    ref = pytest.importorskip("oracle.ref")
    if not ref.available():
        pytest.skip("oracle/_ref not built and no reference tree")
    for P, hw, sm in ((3000, (64, 64), 1.0), (2500, (50, 70), 1.5)):
        c, v = _scene(P, hw, seed=P % 97, sm=sm)
        xyz, rho, sc, q = Hh.cloud_np(c)
        vm, pm = Hh.np_view(v)
        r = ref.raster_forward(xyz, rho, sc, q, 1.0, None, vm, pm, v.tanfovx, v.tanfovy, hw[0], hw[1], v.mode)
        o = Hh.oracle_raster(oracle, c, v)
        dL = S.make_pixel_grad(*hw).numpy()
        gr = ref.raster_backward(r, xyz, sc, q, 1.0, None, vm, pm, v.tanfovx, v.tanfovy, dL)
        st = Pz.raster_grad_parity(oracle, o, dL, gr, xyz, sc, q, 1.0, None, vm, pm, v.tanfovx, v.tanfovy)
        assert st["dL_dmeans3D"]["max_err_over_tol"] < 0.1, st   # float accumulation sits far inside the bound
    # voxelizer
    c = S.make_cloud(3000, seed=5)
    xyz, rho, sc, q = Hh.cloud_np(c)
    n, sv, ctr = (24, 40, 17), (1.5, 2.5, 1.0625), (0.1, -0.2, 0.05)
    r = ref.voxel_forward(xyz, rho, sc, q, 1.0, None, n, sv, ctr)
    o = oracle.voxel_forward(xyz, rho, sc, q, 1.0, None, n, sv, ctr)
    rng = np.random.default_rng(1)
    dL = ((rng.random(n, dtype=np.float32) * 2 - 1) / np.prod(n)).astype(np.float32)
    gr = ref.voxel_backward(r, xyz, sc, q, 1.0, None, dL)
    st = Pz.voxel_grad_parity(oracle, o, dL, gr, sc, q, 1.0, None)
    assert st["dL_dmeans3D"]["max_err_over_tol"] < 0.1, st


def _exp2_image(o):
    """The image evaluated like the HIP forward: log2-domain quadratic form in float32, exp2, same tile lists and tests."""
    H, W = o["H"], o["W"]
    gx = o["grid"][0]
    con, m2, mu = o["conic_opacity"], o["means2D"], o["mus"]
    A2, B2, C2 = (np.float32(-0.5) * LOG2E) * con[:, 0], (-LOG2E) * con[:, 1], (np.float32(-0.5) * LOG2E) * con[:, 2]
    w = con[:, 3] * mu
    with np.errstate(divide="ignore"):
        L = np.where(w > 0, np.log2(w, dtype=np.float32), -np.inf).astype(np.float32)
    img = np.zeros((H, W), np.float32)
    for t in range(o["ranges"].shape[0]):
        r0, r1 = o["ranges"][t]
        if r1 == r0:
            continue
        ids = o["point_list"][r0:r1].astype(np.int64)
        ty, tx = divmod(t, gx)
        ys, xs = np.mgrid[ty * 16:min(ty * 16 + 16, H), tx * 16:min(tx * 16 + 16, W)].astype(np.float32)
        dx = m2[ids, 0, None, None] - xs[None]
        dy = m2[ids, 1, None, None] - ys[None]
        pl = dx * (A2[ids, None, None] * dx + B2[ids, None, None] * dy) + ((C2[ids, None, None] * dy) * dy + L[ids, None, None])
        al = np.exp2(pl, dtype=np.float32)
        ok = (pl <= L[ids, None, None]) & (al >= np.float32(1e-5))
        img[ty * 16:ty * 16 + ys.shape[0], tx * 16:tx * 16 + xs.shape[1]] = np.where(ok, al, np.float32(0)).sum(0, dtype=np.float32)
    return img


def test_exp2_rounding_only_produces_attributed_flips(oracle):
    c, v = _scene(20000, (128, 128), seed=9, sm=0.7)
    o = Hh.oracle_raster(oracle, c, v)
    budget, nb = oracle.raster_forward_audit(o)
    img = _exp2_image(o)
    st = Pz.image_parity(img, o["color"], budget)
    assert st["max_rel_err"] < 1e-4 and st["n_flips"] <= st["n_flip_candidates"] < 0.01 * st["n"], st
    assert int((nb > 0).sum()) == st["n_flip_candidates"]
    # a non-borderline error of 2e-4 relative on one pixel without budget must be caught
    bad = img.copy().reshape(-1)
    i = int(np.argmax((budget.reshape(-1) == 0) & (o["color"].reshape(-1) > 0.01)))
    bad[i] *= np.float32(1.0002)
    with pytest.raises(Pz.ParityError):
        Pz.image_parity(bad, o["color"], budget)
    # ... and a spurious contribution on a pixel whose reference value is exactly zero
    z = np.nonzero((o["color"].reshape(-1) == 0) & (budget.reshape(-1) == 0))[0]
    if z.size:
        bad = img.copy().reshape(-1)
        bad[z[0]] = 1e-5
        with pytest.raises(Pz.ParityError):
            Pz.image_parity(bad, o["color"], budget)


def test_voxel_audit_budget_is_sparse(oracle):
    c = S.make_cloud(5000, seed=4)
    o = Hh.oracle_voxel(oracle, c, (32, 32, 32), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0))
    budget, nb = oracle.voxel_forward_audit(o)
    assert (budget >= 0).all() and (nb > 0).mean() < 0.01
    st = Pz.image_parity(o["vol"], o["vol"], budget, what="volume")
    assert st["n_flips"] == 0


def _raster_case(oracle, P=20000, hw=(128, 128), seed=9, sm=0.7):
    c, v = _scene(P, hw, seed=seed, sm=sm)
    o = Hh.oracle_raster(oracle, c, v)
    dL = S.make_pixel_grad(*hw).numpy()
    xyz, rho, sc, q = Hh.cloud_np(c)
    vm, pm = Hh.np_view(v)
    args = (xyz, sc, q, 1.0, None, vm, pm, v.tanfovx, v.tanfovy)
    return o, dL, args


def _as_boundary(oracle, o, raw, args):
    """Raw sums [P,7] -> the seven arrays the `_C` boundary returns, through the oracle's own geometry chain."""
    ch = oracle.raster_geom_chain(o, raw.astype(np.float32), *args)
    P = raw.shape[0]
    m2 = np.zeros((P, 3), np.float32)
    m2[:, :2] = raw[:, :2]
    return dict(dL_dmeans2D=m2, dL_dopacity=raw[:, 5].astype(np.float32), dL_dmu=raw[:, 6].astype(np.float32),
                dL_dmeans3D=ch["dL_dmeans3D"], dL_dcov3D=ch["dL_dcov3D"], dL_dscales=ch["dL_dscales"],
                dL_drotations=ch["dL_drotations"])


def test_after_flips_closes_the_flagged_row_loophole(oracle):
    """a flagged row's tolerance contains its whole flip budget, so a rounding defect smaller than one
    borderline contribution was invisible there.  With the pair list, (a) a row whose pair REALLY flipped leaves a residual far
    inside the pure tolerance once the flip is taken out, (b) a defect of 3x the pure tolerance planted on a flagged row -- still
    inside tolerance + budget -- is rejected."""
    o, dL, args = _raster_case(oracle)
    s, a, f, pairs = oracle.raster_backward_audit(o, dL, pairs=True)
    assert not pairs["truncated"] and pairs["count"] == len(pairs["ids"]) > 0
    flagged = (f > 0).any(axis=1)
    assert set(np.unique(pairs["ids"])) == set(np.nonzero(flagged)[0])
    # the pair list adds up to the budget the audit books (|terms| x 1.0001 per pair)
    tot = np.zeros_like(f)
    np.add.at(tot, pairs["ids"], np.abs(pairs["vals"]))
    np.testing.assert_allclose(tot * 1.0001, f, rtol=1e-9, atol=1e-300)
    # (0) the exact sums pass with margin 0 after flips
    st = Pz.raster_grad_parity(oracle, o, dL, _as_boundary(oracle, o, s, args), *args)
    af = st["after_flips"]
    assert af["rows_checked"] + af["rows_skipped_many_pairs"] == af["rows_flagged"] > 0
    assert af["max_err_over_tol_after_flips"] < 0.05 and af["rows_needing_a_flip"] == 0, af
    # (a) flip the first listed pair of a few Gaussians for real
    raw = s.copy()
    seen = set()
    for j, i in enumerate(pairs["ids"]):
        if int(i) not in seen and len(seen) < 25:
            seen.add(int(i))
            raw[i] += pairs["vals"][j]
    st = Pz.raster_grad_parity(oracle, o, dL, _as_boundary(oracle, o, raw, args), *args)
    af = st["after_flips"]
    assert af["rows_needing_a_flip"] == len(seen) >= 5 and af["max_err_over_tol_after_flips"] < 0.05, af
    assert st["dL_dopacity"]["max_err_over_tol"] > 0.5        # the old column: near 1.0 by construction
    # (b) a defect of 3x the pure tolerance on the opacity sum of a flagged Gaussian whose budget hides it
    cand = np.nonzero(flagged & (f[:, 5] > 4e-4 * a[:, 5]) & (a[:, 5] > 0))[0]
    assert cand.size
    i = int(cand[0])
    raw = s.copy()
    raw[i, 5] += 3e-4 * a[i, 5]
    g = _as_boundary(oracle, o, raw, args)
    assert abs(float(g["dL_dopacity"][i]) - s[i, 5]) <= 1e-4 * a[i, 5] + f[i, 5]   # inside tolerance + budget ...
    with pytest.raises(Pz.ParityError, match="after taking out"):                   # ... and still caught
        Pz.raster_grad_parity(oracle, o, dL, g, *args)


def test_after_flips_voxel(oracle):
    c = S.make_cloud(6000, seed=4)
    n, sv, ctr = (32, 32, 32), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0)
    o = Hh.oracle_voxel(oracle, c, n, sv, ctr)
    rng = np.random.default_rng(2)
    dL = ((rng.random(n, dtype=np.float32) * 2 - 1) / np.prod(n)).astype(np.float32)
    xyz, rho, sc, q = Hh.cloud_np(c)
    s, a, f, pairs = oracle.voxel_backward_audit(o, dL, pairs=True)
    assert pairs["count"] > 0 and not pairs["truncated"]
    raw = s.copy()
    i0 = int(pairs["ids"][0])
    raw[i0] += pairs["vals"][0]                                   # one real flip
    ch = oracle.voxel_geom_chain(o, raw.astype(np.float32), sc, q, 1.0, None)
    g = dict(dL_dopacity=raw[:, 9].astype(np.float32), dL_dmeans3D=ch["dL_dmeans3D"], dL_dcov3D=ch["dL_dcov3D"],
             dL_dscales=ch["dL_dscales"], dL_drotations=ch["dL_drotations"])
    st = Pz.voxel_grad_parity(oracle, o, dL, g, sc, q, 1.0, None)
    af = st["after_flips"]
    assert af["rows_needing_a_flip"] >= 1 and af["max_err_over_tol_after_flips"] < 0.05, af
