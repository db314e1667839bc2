"""Query time of the voxelizer over a table of (Gaussians, grid) on ONE library (R2HIP_LIB selects it): A/B of a build against
another, run alternately by the calling shell script.   python scripts/voxel_grid_sweep.py [n=20] [cases=5000:64,50000:128,...]"""
import statistics
import sys
import time

import torch

sys.path.insert(0, ".")
from r2_gaussian_amd import _C
from r2_gaussian_amd import scene as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cases = sys.argv[2] if len(sys.argv) > 2 else "5000:64,50000:128,300000:64,300000:128,50000:256,100000:256,300000:256,1000000:256,300000:32"
dev = torch.device("cuda:0")
e = torch.empty(0)
out_line = []
for cs in cases.split(","):
    P, G = (int(x) for x in cs.split(":"))
    c = S.make_cloud(P, seed=0)
    size = 0.25 if G == 32 else 2.0   # 32: the training loop's TV patch
    a = (c.xyz.to(dev), c.density.to(dev), c.scales.to(dev), c.rotations.to(dev), 1.0, e, G, G, G, size, size, size, 0.0, 0.0, 0.0,
         False, False)
    with torch.no_grad():
        for _ in range(3):
            out = _C.voxelize_gaussians(*a)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _j in range(n):
                out = _C.voxelize_gaussians(*a)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / n)
    out_line.append("%dk/%d^3 %.1f" % (P // 1000, G, statistics.median(ts) * 1e6))
print("us/query: " + " | ".join(out_line), flush=True)
