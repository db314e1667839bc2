"""Adaptive density control on the device (csrc/densify_ops.hip, r2_gaussian_amd/densify.py) against the torch statement of
GaussianModel.densify_and_prune / add_densification_stats in tests/mini_trainer.py (which restates
r2_gaussian/gaussian/gaussian_model.py:320-556), with the same split samples: same rows in the same order, parameters and
Adam moments, statistics reset."""
import numpy as np
import pytest
import torch

from tests import mini_trainer as T

pytestmark = pytest.mark.gpu


def _model(gpu, seed=0, n_init=3000):
    case = T.Case(detector=64, n_vol=32, n_views=6, p_gt=1500, n_init=n_init, seed=2)
    opt = T.Opt(iterations=100, densify_scale_threshold=0.02)   # a scale threshold inside the scale range: clones AND splits
    be = T.Backend("hip")
    gen = torch.Generator().manual_seed(seed)
    m = T.Model(case, opt, be, gen)
    g = torch.Generator().manual_seed(seed + 1)
    P = m.P
    # a spread of scales / densities / positions so that every branch is taken: clone, split, prune by density, prune by box
    with torch.no_grad():
        m.p["scaling"].copy_(m.scaling_inv(torch.exp(torch.rand(P, 3, generator=g) * 4.0 - 6.0).clamp(m.lo * 1.01, m.hi * 0.99)).to(gpu))
        m.p["rotation"].copy_(torch.randn(P, 4, generator=g).to(gpu))
        dens = torch.rand(P, 1, generator=g) * 0.3
        dens[::17] = 1.5e-5                                  # halves below the 1e-5 threshold when cloned / split
        dens[::29] = 5e-6                                    # pruned outright
        m.p["density"].copy_(m.inv_softplus(dens).to(gpu))
        xyz = m.p["xyz"].detach().cpu()
        xyz[::23] = xyz[::23] * 0 + torch.tensor([0.995, 0.0, 0.0])   # splits near the box face: some children land outside
        xyz[::31, 1] = 1.2                                   # outside the box
        m.p["xyz"].copy_(xyz.to(gpu))
    # optimizer state + statistics
    for n in m.NAMES:
        m.p[n].grad = torch.randn(m.p[n].shape, generator=g).to(gpu) * 1e-3
    m.optimizer.step()
    m.grad_accum = (torch.rand(P, 1, generator=g) * 3e-4).to(gpu)
    m.denom = torch.randint(0, 4, (P, 1), generator=g).float().to(gpu)      # some zeros: 0/0 -> nan -> 0
    m.max_radii2D = torch.rand(P, generator=g).to(gpu) * 50
    return case, m, g


def _snapshot(m):
    out = {n: m.p[n].detach().cpu().numpy().copy() for n in m.NAMES}
    for grp in m.optimizer.param_groups:
        st = m.optimizer.state[grp["params"][0]]
        out[grp["name"] + ".m"], out[grp["name"] + ".v"] = st["exp_avg"].cpu().numpy().copy(), st["exp_avg_sq"].cpu().numpy().copy()
    out["max_radii2D"], out["grad_accum"], out["denom"] = (t.cpu().numpy().copy() for t in (m.max_radii2D, m.grad_accum, m.denom))
    return out


def test_densify_and_prune_matches_the_torch_statement(gpu):
    case, a, g = _model(gpu)
    _c, b, _g = _model(gpu)
    normals = torch.randn(2, a.P, 3, generator=g)
    P0 = a.P
    a.densify_and_prune(case.bbox, normals_full=normals)
    b.densify_and_prune_fused(case.bbox, normals_full=normals)
    torch.cuda.synchronize()
    A, B = _snapshot(a), _snapshot(b)
    assert A["xyz"].shape == B["xyz"].shape and A["xyz"].shape[0] != P0, (A["xyz"].shape, B["xyz"].shape, P0)
    for k in A:
        assert A[k].shape == B[k].shape, k
        np.testing.assert_allclose(B[k], A[k], rtol=3e-6, atol=1e-7, err_msg=k)
    # rows that are plain copies are bit-identical
    assert np.array_equal(A["rotation"], B["rotation"]) and np.array_equal(A["xyz.m"], B["xyz.m"]) and np.array_equal(A["max_radii2D"], B["max_radii2D"])
    assert not B["grad_accum"].any() and not B["denom"].any()
    # the optimizer still steps on the new parameters
    for n in b.NAMES:
        b.p[n].grad = torch.ones_like(b.p[n])
    b.optimizer.step()


def test_prune_only_when_densification_is_off(gpu):
    case, a, g = _model(gpu, seed=3)
    _c, b, _g = _model(gpu, seed=3)
    a.opt.max_num_gaussians = b.opt.max_num_gaussians = 10       # P >= max: no clone / split, prune only (gaussian_model.py:517-521)
    a.densify_and_prune(case.bbox)
    b.densify_and_prune_fused(case.bbox)
    A, B = _snapshot(a), _snapshot(b)
    assert A["xyz"].shape == B["xyz"].shape and A["xyz"].shape[0] < 3000
    for k in A:
        assert np.array_equal(A[k], B[k]), k


def test_densification_stats_kernel(gpu):
    from r2_gaussian_amd import densify as D
    g = torch.Generator().manual_seed(2)
    P = 5000
    radii = (torch.randint(0, 40, (P,), generator=g) * (torch.rand(P, generator=g) > 0.3)).int().to(gpu)
    g2 = torch.randn(P, 3, generator=g).to(gpu)
    mr, ga, dn = (torch.rand(P, generator=g).to(gpu) * s for s in (60.0, 1.0, 5.0))
    mr0, ga0, dn0 = mr.clone(), ga.clone(), dn.clone()
    D.densification_stats(radii, g2, mr, ga, dn)
    vis = radii > 0
    mr0[vis] = torch.max(mr0[vis], radii[vis].float())
    ga0[vis] += g2[vis, :2].norm(dim=-1)
    dn0[vis] += 1
    assert torch.equal(mr, mr0) and torch.equal(dn, dn0)
    assert torch.allclose(ga, ga0, rtol=1e-6, atol=0)


def test_trainer_with_fused_densify_follows_the_torch_version(gpu):
    case = T.Case(detector=64, n_vol=32, n_views=10, p_gt=2000, n_init=1500, seed=2)
    opt = dict(iterations=160, densify_from_iter=40, densify_until_iter=140, densification_interval=20)
    a = T.train(case, T.Opt(**opt), "hip", eval_every=40, seed=0)
    b = T.train(case, T.Opt(**opt), "hip", eval_every=40, seed=0, fused_losses=True, fused_densify=True)
    assert a["iters"] == b["iters"]
    assert max(abs(x - y) for x, y in zip(a["psnr"], b["psnr"])) < 0.15      # different split samples: not the same run
    assert abs(a["P"][-1] - b["P"][-1]) <= 0.1 * a["P"][-1] and b["P"][-1] > 1500


# ---- against the reference's own GaussianModel.densify_and_prune (tests/golden/train/densify_*.npz, make_golden_train.py):
# A default thresholds, B max_screen_size + max_scale set, C prune only (max_num_gaussians reached)
@pytest.mark.parametrize("case", ["A", "B", "C"])
def test_fused_densify_and_prune_matches_the_reference_fixture(case, gpu):
    import os
    from tests.test_train_golden_cpu import compare_with_fixture, model_from_fixture, snapshot
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "train", "densify_%s.npz" % case))
    m = model_from_fixture(g, T.Backend("hip"))
    P0 = m.P
    m.densify_and_prune_fused(torch.from_numpy(g["bbox"]), normals_full=torch.from_numpy(g["normals"]))
    torch.cuda.synchronize()
    assert m.P == g["out.xyz"].shape[0] and m.P != P0
    compare_with_fixture(snapshot(m), g, exact=True)
    for n in m.NAMES:   # the optimizer still steps on the new parameters
        m.p[n].grad = torch.ones_like(m.p[n])
    m.optimizer.step()


def test_nothing_left_raises_like_the_reference(gpu):
    import os
    from tests.test_train_golden_cpu import model_from_fixture
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "train", "densify_C.npz"))
    m = model_from_fixture(g, T.Backend("hip"))
    m.opt.density_min_threshold = 1e9            # everything is pruned: train.py:169-172 raises
    with pytest.raises(ValueError, match="No Gaussian left"):
        m.densify_and_prune_fused(torch.from_numpy(g["bbox"]))
