"""Only the 256^3 volume query of the benchmark cloud (300k Gaussians), N times: the workload behind `voxelizer.gvoxel_per_s`,
alone, for rocprofv3 (kernel stats / PMC of the voxel kernels without the 32^3 TV-patch launches mixed in)."""
import sys

import torch

sys.path.insert(0, ".")
from r2_gaussian_amd import _C
from r2_gaussian_amd import scene as S

dev = torch.device("cuda:0")
c = S.make_cloud(300000, seed=0)
e = torch.empty(0)
a = (c.xyz.to(dev), c.density.to(dev), c.scales.to(dev), c.rotations.to(dev), 1.0, e, 256, 256, 256, 2.0, 2.0, 2.0, 0.0, 0.0, 0.0,
     False, False)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
with torch.no_grad():
    for _ in range(n):
        R3 = _C.voxelize_gaussians(*a)[0]
torch.cuda.synchronize()
print("R3", R3)
