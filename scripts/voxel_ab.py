"""256^3 query of the benchmark cloud on the general binning chain and on the stick-first chain (csrc/voxel_sticks.hip), alternating
on ONE box: ms per call (median of 5 x n calls, as bench.py times it), per-stage times of the same call, and the two volumes compared
bit for bit.   python scripts/voxel_ab.py [n=20] [P=300000 | small | large] [grid=256] [modes=0,1] [nolimit]   (r2_voxel_sticks_control modes)"""
import ctypes as C
import statistics
import sys
import time

import torch

sys.path.insert(0, ".")
from r2_gaussian_amd import _C, _lib
from r2_gaussian_amd import scene as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
CLOUD = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].isdigit() else None   # "small" / "large": a trained cloud (tests/trained_cloud.py)
P = int(sys.argv[2]) if len(sys.argv) > 2 and CLOUD is None else 300000
G = int(sys.argv[3]) if len(sys.argv) > 3 else 256
MODES = tuple(int(x) for x in sys.argv[4].split(",")) if len(sys.argv) > 4 else (0, 1)
dev = torch.device("cuda:0")
if CLOUD:
    from tests import trained_cloud as TCl
    c, info = TCl.load(CLOUD)
    P = c.xyz.shape[0]
    print("trained cloud %s: %d Gaussians" % (CLOUD, P), flush=True)
else:
    c = S.make_cloud(P, seed=0)
e = torch.empty(0)
a = (c.xyz.to(dev), c.density.to(dev), c.scales.to(dev), c.rotations.to(dev), 1.0, e, G, G, G, 2.0, 2.0, 2.0, 0.0, 0.0, 0.0,
     False, False)
L = _lib.lib()
if len(sys.argv) > 5 and sys.argv[5] == "nolimit":   # keep scenes with very long lists on the stick chain (r2_voxel_sticks_limits)
    L.r2_voxel_sticks_limits(C.c_longlong(1 << 40), C.c_longlong(1 << 40))
vols = {}
for rep in range(2):
    for mode in MODES:
        L.r2_voxel_sticks_control(mode)
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
            _lib.profile_enable(None)
            for _ in range(5):
                _C.voxelize_gaussians(*a)
            torch.cuda.synchronize()
            prof = _lib.profile_read(reset=True)
            _lib.profile_enable([])
        st = (C.c_longlong * 3)()
        L.r2_voxel_sticks_stats(st, 1)
        tv = statistics.median(ts)
        print("mode %d rep %d: %.1f us/call (min %.1f) = %.2f GVoxel/s, R %d, sticks taken/fallback/declined %s; %s" % (
            mode, rep, tv * 1e6, min(ts) * 1e6, G ** 3 / tv / 1e9, out[0], list(st),
            " ".join("%s %.1f" % (k.replace("voxel.", ""), 1e3 * ms / cnt) for k, (ms, cnt) in sorted(prof.items()) if k.startswith("voxel."))),
            flush=True)
        vols[mode] = out[1].clone()
L.r2_voxel_sticks_control(1)
same = torch.equal(vols[MODES[0]].view(torch.int32), vols[MODES[-1]].view(torch.int32))
print("volumes identical bit for bit:", same)
sys.exit(0 if same else 1)
