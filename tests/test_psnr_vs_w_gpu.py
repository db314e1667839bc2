"""GPU: the 3D PSNR survives view-sharded training (SURVEY.md 8e / 7 hard part 5).

The reference takes one view per optimiser step (train.py:104-113); N view-sharded ranks take W = N.  tests/mini_trainer
renders the W views of a step on one GPU, bit-identically to the world-W run (tests/test_dist_cpu.py), so the optimiser
question is settled here, at a reduced size (128^2 detector, 64^3 volume; the full-size study is scripts/psnr_vs_w.py ->
profiles/r05_psnr_vs_W.json, table in DESIGN.md section 6):
  * equal optimiser steps, W = 8 views per step, the reference's schedule unchanged: PSNR must not fall (it rises);
  * equal VIEWS: 1/8 of the steps, densification window / interval and LR horizon divided by 8, all four learning rates x 8
    (the configuration DESIGN.md section 6 states for N = 8): final 3D PSNR within 0.1 dB of W = 1, or better.
"""
import pytest

pytestmark = pytest.mark.gpu


def test_eight_views_per_step_keep_the_psnr(gpu):
    from scripts.psnr_vs_w import make_opt
    from tests import mini_trainer as T
    case = T.Case(detector=128, n_vol=64, n_views=50, p_gt=20000, n_init=5000, seed=2)
    sched = (1200, 200, 700, 50)
    extra = dict(densify_grad_threshold=5e-5, max_num_gaussians=300000)

    def final(W, family, rule):
        opt = make_opt(T, *sched, W, family, rule, extra)
        out = T.train(case, opt, "hip", eval_every=opt.iterations, seed=0, fused_losses=True, fused_densify=True, views_per_step=W)
        return out["psnr"][-1]

    base = final(1, "steps", "1")
    assert base > 25.0                                   # it trains (measured: 26.35 dB, run-to-run spread 0.01 dB)
    same_steps = final(8, "steps", "1")
    same_views = final(8, "views", "lin")
    assert same_steps >= base - 0.1, (base, same_steps)  # measured: 29.04 dB
    assert same_views >= base - 0.1, (base, same_views)  # measured: 26.80 dB
