"""Generates tests/golden/psnr_traj_oracle.json: the 3D-PSNR trajectory of the miniature trainer (tests/mini_trainer.py) on
the synthetic cone-beam case with the CPU ORACLE as renderer / voxelizer -- the reference's arithmetic end to end.
tests/test_psnr_parity_gpu.py trains the same case with the HIP kernels and compares at matched iterations (<= 0.1 dB).

    python tests/golden/make_psnr_traj.py            (about 10 minutes on 8 cores)
    python tests/golden/make_psnr_traj.py medium     (256^2 detector / 128^3 volume / 50k-Gaussian phantom; about an hour)
    python tests/golden/make_psnr_traj.py large      (512^2 detector / 256^3 volume / 150k-Gaussian phantom; 1-2 hours on 8 cores)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CASE = dict(detector=128, n_vol=64, n_views=50, p_gt=20000, n_init=5000, seed=2)
OPT = dict(iterations=1500, densify_from_iter=300, densify_until_iter=1200, densification_interval=100)
EVAL_EVERY = 100
MEDIUM = dict(case=dict(detector=256, n_vol=128, n_views=50, p_gt=50000, n_init=12000, seed=3),
              opt=dict(iterations=2000, densify_from_iter=300, densify_until_iter=1500, densification_interval=100),
              eval_every=200, file="psnr_traj_oracle_medium.json")
# round 4: the headline's detector and query volume (512^2 / 256^3), a 150k-Gaussian phantom, 40k -> ~150k Gaussians
LARGE = dict(case=dict(detector=512, n_vol=256, n_views=50, p_gt=150000, n_init=40000, seed=4),
             opt=dict(iterations=1200, densify_from_iter=200, densify_until_iter=900, densification_interval=100),
             eval_every=200, file="psnr_traj_oracle_large.json")

if __name__ == "__main__":
    import torch
    from tests import mini_trainer as T
    torch.set_num_threads(int(os.environ.get("R2_GOLDEN_THREADS", os.cpu_count())))
    name = "psnr_traj_oracle.json"
    if len(sys.argv) > 1 and sys.argv[1] in ("medium", "large"):
        PRESET = MEDIUM if sys.argv[1] == "medium" else LARGE
        CASE, OPT, EVAL_EVERY, name = PRESET["case"], PRESET["opt"], PRESET["eval_every"], PRESET["file"]
    case = T.Case(**CASE)
    out = T.train(case, T.Opt(**OPT), "oracle", eval_every=EVAL_EVERY, seed=0, log=print)
    out.update(case=CASE, opt=OPT, eval_every=EVAL_EVERY)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), name), "w") as f:
        json.dump(out, f, indent=1)
    print("final psnr3d %.3f dB, P %d, %.2f it/s" % (out["psnr"][-1], out["P"][-1], out["it_per_s"]))
