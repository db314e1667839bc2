"""Parity accounting against the CPU oracle (TEST INFRASTRUCTURE ONLY -- same rules as oracle.py: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline / self-check leg import this).

north_star's bar is "rendered projections and voxel grids within 1e-4 rel".  The reference's render kernels contain two
discontinuous tests per (pixel, Gaussian) pair (power > 0, alpha < cut-off: RAS/forward.cu:369-375, VOX/forward.cu:274-293),
so an implementation that rounds alpha differently can take the other branch for a pair that sits ON a threshold.  Instead
of hiding those under an absolute floor, the oracle's audit functions (r2_oracle.c, AUDIT section) book for every output
element how much of it hangs on borderline pairs; the checks here assert

    |got - ref| <= rtol * |ref| + flip_budget                                   (images, volumes)
    |got - ref| <= rtol * sum|terms| + flip_budget                              (the raw gradient sums of the render backward)
    |got - ref| <= |J| (rtol * sum|terms| + flip_budget) + 1e-6 |J| |raw|       (gradients behind the geometry chain, J =
                                                                                 the reference's own float-evaluated Jacobian)

with rtol = 1e-4 and nothing else, and report how many elements carry a budget and how many needed it.  sum|terms| is the
scale a float32 sum with cancellation is accurate to (the reference itself accumulates these sums with float atomics in
arbitrary order): a plain |ref|-relative bound is meaningless for a sum that cancels to ~0.
"""
import numpy as np

RTOL = 1e-4


class ParityError(AssertionError):
    pass


def image_parity(got, ref, budget, rtol=RTOL, what="image"):
    """-> stats dict; raises ParityError when an element is outside rtol*|ref| + budget."""
    got = np.asarray(got, np.float64).reshape(-1)
    ref = np.asarray(ref, np.float64).reshape(-1)
    budget = np.asarray(budget, np.float64).reshape(-1)
    assert got.shape == ref.shape == budget.shape, (got.shape, ref.shape, budget.shape)
    err = np.abs(got - ref)
    pure = rtol * np.abs(ref)
    clean = budget == 0.0
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.where(ref != 0.0, err / np.abs(ref), np.where(err == 0.0, 0.0, np.inf))
    stats = {
        "n": int(got.size),
        "max_rel_err": float(rel[clean].max()) if clean.any() else 0.0,      # over elements WITHOUT any borderline pair
        "n_flip_candidates": int((~clean).sum()),                             # elements holding >= 1 borderline pair
        "n_flips": int((err > pure).sum()),                                   # elements that actually needed their budget
        "max_flip_excess": float((err - pure)[~clean].max()) if (~clean).any() else 0.0,
    }
    bad = err > pure + budget
    stats["n_bad"] = int(bad.sum())
    if bad.any():
        i = int(np.argmax(err - pure - budget))
        raise ParityError("%s: %d / %d elements outside %.0e*|ref| + flip budget; worst at %d: got %.9g ref %.9g budget %.3g (%s)"
                          % (what, bad.sum(), got.size, rtol, i, got[i], ref[i], budget[i], stats))
    return stats


def _sum_check(name, got, ref64, tol, stats, flagged):
    got = np.asarray(got, np.float64).reshape(ref64.shape)
    err = np.abs(got - ref64)
    bad = err > tol
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.where(tol > 0, err / tol, np.where(err == 0, 0.0, np.inf))
    scale = float(np.abs(ref64).max()) if ref64.size else 0.0
    unfl = ~flagged if flagged.ndim == err.ndim else ~flagged.reshape((-1,) + (1,) * (err.ndim - 1))
    unfl = np.broadcast_to(unfl, err.shape)
    stats[name] = {
        "max_err_over_tol": float(ratio.max()) if ratio.size else 0.0,
        # ... over the rows WITHOUT a borderline pair: the margin of the arithmetic itself.  (A row whose pair really flipped
        # differs by that pair's whole contribution, which is what its budget holds: such rows sit just below 1 by
        # construction and say nothing about rounding.)
        "max_err_over_tol_unflagged": float(ratio[unfl].max()) if unfl.any() else 0.0,
        "n_flagged_rows": int(flagged.sum()),
        "max_err_over_scale_unflagged": float(err[unfl].max() / scale) if scale > 0 and unfl.any() else 0.0,
        "n_bad": int(bad.sum()),
    }
    if bad.any():   # collected; the caller raises once every array has been looked at (the report shows all margins)
        i = np.unravel_index(int(np.argmax(ratio)), err.shape)
        stats.setdefault("_errors", []).append(
            "%s: %d / %d elements out of tolerance; worst at %s: got %.9g ref %.9g tol %.3g (scale %.3g)"
            % (name, bad.sum(), err.size, i, got[i], ref64[i], tol[i], scale))


def _finish(stats, what):
    if stats.get("_errors"):
        public = {k: v for k, v in stats.items() if not k.startswith("_")}
        raise ParityError("%s: %s\n%s" % (what, "; ".join(stats["_errors"]), public))
    return stats


def raster_grad_parity(O, st, dL, gh, means3D, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
                       tanfovx, tanfovy, rtol=RTOL):
    """HIP gradients `gh` (names as returned by the `_C` mirror) vs the oracle's double-accumulated sums pushed through the
    reference's geometry chain.  -> stats dict; raises ParityError."""
    P = st["P"]
    s, a, f = O.raster_backward_audit(st, dL)
    tol_raw = rtol * a + f
    flagged = (f > 0).any(axis=1)
    stats = {"P": int(P), "n_flip_candidates": int(flagged.sum()), "_flagged": flagged}
    # raw sums that the boundary returns as they are
    _sum_check("dL_dmeans2D", np.asarray(gh["dL_dmeans2D"])[:, 0:2], s[:, 0:2], tol_raw[:, 0:2], stats, flagged)
    assert not np.asarray(gh["dL_dmeans2D"])[:, 2].any()   # RAS/backward.cu never writes the third component (Q13)
    _sum_check("dL_dopacity", np.asarray(gh["dL_dopacity"]).reshape(P), s[:, 5], tol_raw[:, 5], stats, flagged)
    _sum_check("dL_dmu", np.asarray(gh["dL_dmu"]).reshape(P), s[:, 6], tol_raw[:, 6], stats, flagged)
    # gradients behind the geometry chain: reference value = chain(raw sums), tolerance = |J| tol_raw
    args = (means3D, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tanfovx, tanfovy)
    ref = O.raster_geom_chain(st, s.astype(np.float32), *args)
    finals = ["dL_dmeans3D", "dL_dcov3D"] + ([] if cov3D_precomp is not None else ["dL_dscales", "dL_drotations"])
    tol = {k: np.zeros(ref[k].shape, np.float64) for k in finals}
    for q in (0, 1, 2, 3, 4, 6):   # opacity (5) feeds nothing downstream
        unit = np.zeros((P, 7), np.float32)
        unit[:, q] = 1.0
        J = O.raster_geom_chain(st, unit, *args)
        for k in finals:
            Jk = np.abs(J[k].astype(np.float64))
            tol[k] += Jk * (tol_raw[:, q:q + 1] + 1e-6 * np.abs(s[:, q:q + 1]))
    for k in finals:
        _sum_check(k, gh[k], ref[k].astype(np.float64), tol[k], stats, flagged)
    return _finish(stats, "rasterizer gradients")


def voxel_grad_parity(O, st, dL, gh, scales, rotations, scale_modifier, cov3D_precomp, rtol=RTOL):
    P = st["P"]
    s, a, f = O.voxel_backward_audit(st, dL)
    tol_raw = rtol * a + f
    flagged = (f > 0).any(axis=1)
    stats = {"P": int(P), "n_flip_candidates": int(flagged.sum()), "_flagged": flagged}
    _sum_check("dL_dopacity", np.asarray(gh["dL_dopacity"]).reshape(P), s[:, 9], tol_raw[:, 9], stats, flagged)
    ref = O.voxel_geom_chain(st, s.astype(np.float32), scales, rotations, scale_modifier, cov3D_precomp)
    finals = ["dL_dmeans3D", "dL_dcov3D"] + ([] if cov3D_precomp is not None else ["dL_dscales", "dL_drotations"])
    tol = {k: np.zeros(ref[k].shape, np.float64) for k in finals}
    for q in range(9):
        unit = np.zeros((P, 10), np.float32)
        unit[:, q] = 1.0
        J = O.voxel_geom_chain(st, unit, scales, rotations, scale_modifier, cov3D_precomp)
        for k in finals:
            tol[k] += np.abs(J[k].astype(np.float64)) * (tol_raw[:, q:q + 1] + 1e-6 * np.abs(s[:, q:q + 1]))
    for k in finals:
        _sum_check(k, gh[k], ref[k].astype(np.float64), tol[k], stats, flagged)
    return _finish(stats, "voxelizer gradients")
