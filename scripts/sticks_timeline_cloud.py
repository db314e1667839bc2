"""In-kernel stamps of the voxelizer's stick-first chain on a given cloud (a trained one or a synthetic one of P Gaussians), through the
ctypes boundary of the stamped experiment build:
    R2HIP_LIB=r2_gaussian_amd/libr2hip_ts.so python scripts/sticks_timeline_cloud.py small|large|P out.bin"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from r2_gaussian_amd import _C, _lib
from r2_gaussian_amd import scene as S

what, out = sys.argv[1], sys.argv[2]
if what.isdigit():
    c = S.make_cloud(int(what), seed=0)
else:
    from tests import trained_cloud as TCl
    c, _info = TCl.load(what)
dev = torch.device("cuda:0")
e = torch.empty(0)
a = (c.xyz.to(dev), c.density.to(dev), c.scales.to(dev), c.rotations.to(dev), 1.0, e, 256, 256, 256, 2.0, 2.0, 2.0, 0.0, 0.0, 0.0,
     False, False)
with torch.no_grad():
    for _ in range(4):
        r = _C.voxelize_gaussians(*a)
torch.cuda.synchronize()
with torch.no_grad():
    r = _C.voxelize_gaussians(*a)
torch.cuda.synchronize()
ts = (C.c_ulonglong * (16 * 2048))()
rc = _lib.lib().r2_debug_ts_sticks(ts)
np.frombuffer(ts, np.uint64).tofile(out)
print("cloud %s: P %d R %d, stamps -> %s (rc %d)" % (what, c.xyz.shape[0], r[0], out, rc))
