"""3D PSNR against the number of views per optimiser step (SURVEY.md 7 hard part 5, 8e).

The reference takes ONE view per optimiser step (train.py:104-113).  View-sharded data parallelism on N GPUs makes it W = N
(or 2 N with two views per rank): the same model, a different optimiser.  tests/mini_trainer.train(views_per_step=W) renders
the W views of a step one after the other on ONE GPU and is bit-identical to the world-W run over torch.distributed
(tests/test_dist_cpu.py), so the whole question -- does the PSNR survive, and under which step-count / learning-rate rule --
can be answered on one GPU.  Two families of runs on the HIP backend:

  steps   equal optimiser steps: the reference's schedule (iterations, densification window, LR decay) unchanged, W views per
          step, gradients averaged.  Costs W x the views; on W GPUs it costs the same wall time as W = 1 on one.
  views   equal views processed: iterations, densification window and interval and the LR decay horizon all divided by W, so a
          run sees the same number of projections as W = 1 and finishes W x sooner on W GPUs.  Learning-rate rule `lr_rule`:
          "1" unchanged, "sqrt" all four group rates x sqrt(W), "lin" x W (init and final alike).

Every run reports the 3D PSNR trajectory (every `eval_views` views), the Gaussian counts and the single-GPU iterations/s;
W = 1 is repeated over seeds to show the run-to-run spread that any comparison has to be read against.

    python scripts/psnr_vs_w.py [--detector 512 --nvol 256 --iterations 3000] [--ws 2,8,16] [--out gpurun_out/psnr_vs_W.json]
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LR_KEYS = ("position_lr", "density_lr", "scaling_lr", "rotation_lr")


def make_opt(T, n, from_it, until_it, interval, W, family, lr_rule, extra):
    kw = dict(extra)
    if family == "steps":
        kw.update(iterations=n, densify_from_iter=from_it, densify_until_iter=until_it, densification_interval=interval)
    else:
        kw.update(iterations=max(1, n // W), densify_from_iter=max(1, from_it // W), densify_until_iter=max(2, until_it // W),
                  densification_interval=max(1, interval // W))
        f = {"1": 1.0, "sqrt": math.sqrt(W), "lin": float(W)}[lr_rule]
        for k in LR_KEYS:
            kw[k + "_init"] = getattr(T.Opt, k + "_init") * f
            kw[k + "_final"] = getattr(T.Opt, k + "_final") * f
    return T.Opt(**kw)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--detector", type=int, default=512)
    ap.add_argument("--nvol", type=int, default=256)
    ap.add_argument("--views", type=int, default=50)
    ap.add_argument("--pgt", type=int, default=20000)
    ap.add_argument("--init", type=int, default=50000)
    ap.add_argument("--iterations", type=int, default=3000)
    ap.add_argument("--densify-from", type=int, default=500)
    ap.add_argument("--densify-until", type=int, default=1800)
    ap.add_argument("--interval", type=int, default=100)
    ap.add_argument("--grad-threshold", type=float, default=1.2e-5)
    ap.add_argument("--max-gaussians", type=int, default=300000)
    ap.add_argument("--ws", default="2,8,16")
    ap.add_argument("--seeds", default="0,1,2", help="seeds of the W = 1 baseline (run-to-run spread)")
    ap.add_argument("--families", default="steps,views")
    ap.add_argument("--lr-rules", default="1,sqrt,lin")
    ap.add_argument("--eval-views", type=int, default=200, help="3D PSNR every this many processed views (W = 1: iterations)")
    ap.add_argument("--backend", default="hip", help="hip (the product) | oracle (CPU; smoke-testing this script at toy sizes)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "psnr_vs_W.json"))
    a = ap.parse_args()

    import torch
    from tests import mini_trainer as T
    t0 = time.time()
    case = T.Case(detector=a.detector, n_vol=a.nvol, n_views=a.views, p_gt=a.pgt, n_init=a.init, seed=2)
    res = dict(case=dict(detector=a.detector, n_vol=a.nvol, n_views=a.views, p_gt=a.pgt, n_init=a.init),
               schedule=dict(iterations=a.iterations, densify_from=a.densify_from, densify_until=a.densify_until,
                             interval=a.interval, grad_threshold=a.grad_threshold, max_gaussians=a.max_gaussians),
               gt_build_s=round(time.time() - t0, 1), backend=a.backend,
               device=torch.cuda.get_device_name(0) if a.backend == "hip" else "cpu", runs=[])
    extra = dict(densify_grad_threshold=a.grad_threshold, max_num_gaussians=a.max_gaussians)

    def run(W, family, lr_rule, seed):
        opt = make_opt(T, a.iterations, a.densify_from, a.densify_until, a.interval, W, family, lr_rule, extra)
        ev = max(1, a.eval_views // (W if family == "views" else 1))
        t1 = time.time()
        hip = a.backend == "hip"
        out = T.train(case, opt, a.backend, eval_every=ev, seed=seed, fused_losses=hip, fused_densify=hip, views_per_step=W)
        r = dict(W=W, family=family, lr_rule=lr_rule, seed=seed, steps=opt.iterations, views=opt.iterations * W,
                 iters=out["iters"], views_at=[i * W for i in out["iters"]], psnr=[round(p, 4) for p in out["psnr"]], P=out["P"],
                 final_psnr=round(out["psnr"][-1], 4), final_P=out["P"][-1], it_per_s=round(out["it_per_s"], 1),
                 wall_s=round(time.time() - t1, 1))
        res["runs"].append(r)
        print("W %2d %-5s lr %-4s seed %d: %5d steps %6d views -> %.3f dB, P %d, %.1f it/s (%.0f s)" % (
            W, family, lr_rule, seed, r["steps"], r["views"], r["final_psnr"], r["final_P"], r["it_per_s"], r["wall_s"]), flush=True)
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)

    for sd in [int(x) for x in a.seeds.split(",")]:
        run(1, "steps", "1", sd)
    ws = [int(x) for x in a.ws.split(",") if x]
    for fam in a.families.split(","):
        for W in ws:
            if fam == "steps":
                run(W, "steps", "1", 0)
            else:
                for rule in a.lr_rules.split(","):
                    run(W, "views", rule, 0)
    base = [r["final_psnr"] for r in res["runs"] if r["W"] == 1]
    res["baseline_mean"] = sum(base) / len(base)
    res["baseline_spread"] = max(base) - min(base)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    print("W = 1 baseline: %.3f dB mean, spread %.3f dB over %d seeds" % (res["baseline_mean"], res["baseline_spread"], len(base)))
