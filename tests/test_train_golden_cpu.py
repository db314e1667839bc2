"""The repo's restatement of the reference's training-side Python (tests/mini_trainer.py: loss stack, learning-rate schedules,
Adam groups, densify_and_prune with the optimizer-state surgery) against fixtures written by the REFERENCE's own code
(tests/golden/train/*.npz, generator tests/golden/make_golden_train.py: loss_utils.py:19-104, gaussian_utils.py:13-46,
gaussian_model.py:188-235,380-550).  CPU only; the HIP operators are checked against the same files in
tests/test_losses_gpu.py / tests/test_densify_gpu.py."""
import os

import numpy as np
import pytest
import torch

from tests import mini_trainer as T

GOLD = os.path.join(os.path.dirname(__file__), "golden", "train")


def _load(name):
    return np.load(os.path.join(GOLD, name))


def test_losses_match_the_reference_values_and_gradients():
    g = _load("losses.npz")
    for i in range(3):
        gt = torch.from_numpy(g["gt%d" % i])
        img = torch.from_numpy(g["img%d" % i]).requires_grad_(True)
        l1 = (img - gt).abs().mean()
        (gl1,) = torch.autograd.grad(l1, img)
        assert float(l1.detach()) == float(g["l1_%d" % i])
        np.testing.assert_array_equal(gl1.numpy(), g["l1_grad%d" % i])
        s = T.ssim(img, gt)
        (gs,) = torch.autograd.grad(s, img)
        np.testing.assert_allclose(float(s), float(g["ssim_%d" % i]), rtol=2e-6)
        np.testing.assert_allclose(gs.numpy(), g["ssim_grad%d" % i], rtol=0, atol=2e-6 * np.abs(g["ssim_grad%d" % i]).max())
    for i in range(2):
        vol = torch.from_numpy(g["vol%d" % i]).requires_grad_(True)
        tv = T.tv3d_mean(vol)
        (gtv,) = torch.autograd.grad(tv, vol)
        np.testing.assert_allclose(float(tv), float(g["tv_%d" % i]), rtol=1e-6)
        np.testing.assert_allclose(gtv.numpy(), g["tv_grad%d" % i], rtol=0, atol=1e-7 * np.abs(g["tv_grad%d" % i]).max() + 1e-12)


def test_learning_rate_schedules_and_adam_groups():
    g = _load("lr.npz")
    opt = T.Opt(iterations=30000)
    sched = {"xyz": T.expon_lr(opt.position_lr_init, opt.position_lr_final, 30000),
             "density": T.expon_lr(opt.density_lr_init, opt.density_lr_final, 30000),
             "scaling": T.expon_lr(opt.scaling_lr_init, opt.scaling_lr_final, 30000),
             "rotation": T.expon_lr(opt.rotation_lr_init, opt.rotation_lr_final, 30000)}
    for name, f in sched.items():
        ours = np.array([f(int(s)) for s in g["steps"]])
        np.testing.assert_allclose(ours, g["lr_" + name], rtol=1e-14)
    assert T.expon_lr(0.0, 0.0, 10)(3) == float(g["disabled"][0]) == 0.0
    assert T.expon_lr(1e-3, 1e-4, 10)(-1) == float(g["disabled"][1]) == 0.0
    # optimizer: four groups in the reference's order, their initial rates, Adam eps 1e-15 / default betas
    raw = {"xyz": torch.zeros(4, 3), "density": torch.zeros(4, 1), "scaling": torch.zeros(4, 3), "rotation": torch.zeros(4, 4)}
    m = T.Model.from_tensors(opt, T.Backend("oracle"), raw)
    assert [grp["name"] for grp in m.optimizer.param_groups] == list(g["group_names"])
    np.testing.assert_allclose([grp["lr"] for grp in m.optimizer.param_groups], g["group_lr0"], rtol=1e-14)
    assert all(grp["eps"] == e for grp, e in zip(m.optimizer.param_groups, g["eps"]))
    assert all(tuple(grp["betas"]) == tuple(b) for grp, b in zip(m.optimizer.param_groups, g["betas"]))


def model_from_fixture(g, backend, device=None):
    """mini_trainer.Model carrying the fixture's inputs and thresholds."""
    cfg = g["cfg"]   # max_grad, min_density, max_screen_size (0 = None), max_scale (0 = None), max_num_gaussians, scale_thr
    sb = g["scale_bound"]
    opt = T.Opt(iterations=30000, densify_grad_threshold=float(cfg[0]), density_min_threshold=float(cfg[1]),
                max_screen_size=float(cfg[2]) or None, max_scale=(float(cfg[3]) / 2.0) or None,
                max_num_gaussians=int(cfg[4]), densify_scale_threshold=float(cfg[5]) / 2.0,
                scale_min=float(sb[0]) / 2.0, scale_max=float(sb[1]) / 2.0)   # the trainer multiplies by volume_to_world = 2
    t = lambda k: torch.from_numpy(g[k])
    raw = {n: t("in." + n) for n in T.Model.NAMES}
    moments = {n: (t("in.%s.m" % n), t("in.%s.v" % n)) for n in T.Model.NAMES}
    steps = {n: float(g["in.%s.step" % n]) for n in T.Model.NAMES}
    return T.Model.from_tensors(opt, backend, raw, moments, steps, t("in.max_radii2D"), t("in.grad_accum"), t("in.denom"))


def snapshot(m):
    out = {n: m.p[n].detach().cpu().numpy() for n in m.NAMES}
    for grp in m.optimizer.param_groups:
        st = m.optimizer.state[grp["params"][0]]
        out[grp["name"] + ".m"], out[grp["name"] + ".v"] = st["exp_avg"].cpu().numpy(), st["exp_avg_sq"].cpu().numpy()
        out[grp["name"] + ".step"] = np.asarray(float(st["step"]))
    out["max_radii2D"], out["grad_accum"], out["denom"] = (x.cpu().numpy() for x in (m.max_radii2D, m.grad_accum, m.denom))
    return out


def compare_with_fixture(got, g, exact):
    keys = [k[4:] for k in g.files if k.startswith("out.")]
    assert keys
    for k in keys:
        ref = g["out." + k]
        assert got[k].shape == ref.shape, (k, got[k].shape, ref.shape)
        # rows that are copies (rotations, Adam moments, statistics, steps) are bit-identical; computed values (children's
        # positions through the rotation matrix, halved densities, shrunk scales) agree to float rounding
        if exact and k not in ("xyz", "density", "scaling"):
            np.testing.assert_array_equal(got[k], ref, err_msg=k)
        else:
            np.testing.assert_allclose(got[k], ref, rtol=3e-6, atol=1e-7, err_msg=k)


@pytest.mark.parametrize("case", ["A", "B", "C"])
def test_densify_and_prune_restatement_equals_the_reference(case):
    g = _load("densify_%s.npz" % case)
    m = model_from_fixture(g, T.Backend("oracle"))
    P0 = m.P
    m.densify_and_prune(torch.from_numpy(g["bbox"]), normals_full=torch.from_numpy(g["normals"]))
    assert m.P == g["out.xyz"].shape[0] and m.P != P0
    compare_with_fixture(snapshot(m), g, exact=True)
