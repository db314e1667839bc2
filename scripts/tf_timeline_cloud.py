"""In-kernel stamps of the tile-first binning kernels (scatter: phases 0-2, sort: 3 start / 6-9 inside the multi-part path / 4 end)
on a trained cloud: `R2HIP_LIB=.../libr2hip_ts.so python scripts/tf_timeline_cloud.py [small|large|synthetic] [view]`
(needs the -DR2_EXP_TS build: python -m r2_gaussian_amd.build -DR2_EXP_TS --out=libr2hip_ts.so)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from r2_gaussian_amd import _C, _lib, scene as S   # noqa: E402
from tests import trained_cloud as TC   # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "large"
    vi = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dev = torch.device("cuda:0")
    c = S.make_cloud(300000, seed=0) if name == "synthetic" else TC.load(name, train=True)[0]
    v = S.make_views(50, (512, 512))[vi]
    e = torch.empty(0)
    args = (c.xyz.to(dev), c.density.to(dev), c.scales.to(dev), c.rotations.to(dev), 1.0, e, v.world_view_transform.to(dev),
            v.full_proj_transform.to(dev), v.tanfovx, v.tanfovy, v.image_height, v.image_width, v.camera_center.to(dev), False,
            v.mode, False)
    L = _lib.lib()
    for _ in range(6):
        R = _C.rasterize_gaussians(*args)[0]
    torch.cuda.synchronize()
    print("cloud %s P %d view %d R %d" % (name, c.xyz.shape[0], vi, R))
    t0 = None
    rows = []
    for unit in ("geom", "tilefirst", "render"):
        try:
            f = getattr(L, "r2_debug_ts_" + unit)
        except AttributeError:
            print("no stamps for", unit, "(not a -DR2_EXP_TS build)")
            continue
        f.restype = C.c_int
        buf = (C.c_ulonglong * (16 * 2048))()
        f(buf)
        a = np.frombuffer(buf, dtype=np.uint64).reshape(16, 2048).astype(np.float64)
        rows.append((unit, a))
        nz = a[a > 0]
        if nz.size:
            t0 = nz.min() if t0 is None else min(t0, nz.min())
    for unit, a in rows:
        for ph in range(16):
            x = a[ph][a[ph] > 0]
            if x.size:
                x = np.sort((x - t0) * 0.01)
                print("  TS %-9s %2d: n %4d  min %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f us" % (unit, ph, x.size, x[0], x[x.size // 2],
                                                                                           x[int(x.size * 0.9)], x[-1]))


if __name__ == "__main__":
    main()
