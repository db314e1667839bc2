"""HIP-only training run on the synthetic cone-beam case at the headline sizes (512^2 detector, 256^3 volume, 50 views):
iterations/s of the whole training iteration (render + TV query + losses + backward + Adam + densify, all on the GPU through
the drop-in packages) and the final 3D PSNR.  Results -> gpurun_out/train_synthetic.json (copied to profiles/).

    python scripts/train_synthetic.py [--iterations 3000] [--detector 512] [--nvol 256]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=3000)
    ap.add_argument("--detector", type=int, default=512)
    ap.add_argument("--nvol", type=int, default=256)
    ap.add_argument("--init", type=int, default=50000)
    ap.add_argument("--fused-losses", action="store_true", help="loss stack on the HIP kernels (r2_gaussian_amd.losses)")
    args = ap.parse_args()
    from tests import mini_trainer as T
    t0 = time.time()
    case = T.Case(detector=args.detector, n_vol=args.nvol, n_views=50, p_gt=20000, n_init=args.init, seed=2)
    t_case = time.time() - t0
    n = args.iterations
    opt = T.Opt(iterations=n, densify_from_iter=n // 6, densify_until_iter=n // 2, densification_interval=100)
    out = T.train(case, opt, "hip", eval_every=max(100, n // 10), seed=0, log=print, fused_losses=args.fused_losses,
                  fused_densify=args.fused_losses)
    out["fused_losses"] = bool(args.fused_losses)
    out.update(detector=args.detector, n_vol=args.nvol, init=args.init, gt_build_s=round(t_case, 1),
               note="whole training iteration incl. losses / Adam / densify in torch; GT projections + volume by the CPU oracle")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "train_synthetic%s.json" % ("_fused" if args.fused_losses else "")), "w") as f:
        json.dump(out, f, indent=1)
    print("final psnr3d %.3f dB, P %d, %.1f it/s" % (out["psnr"][-1], out["P"][-1], out["it_per_s"]))
