"""Timeline of one training view driven through the Python drop-in layer, from the s_memrealtime stamps of an experiment build
(python -m r2_gaussian_amd.build -DR2_EXP_TS --out=libr2hip_ts.so; R2HIP_LIB=.../libr2hip_ts.so R2_SHIM=0 python scripts/ts_python.py).
Shows where the GPU idles while the host reacts to num_rendered (compare with scripts/cbench on the same library)."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from r2_gaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib
from r2_gaussian_amd import scene as S

dev = torch.device("cuda:0")
c = S.make_cloud(300000, seed=0)
views = S.make_views(50, (512, 512))
xyz, dens, sc, rot = (t.to(dev).requires_grad_(True) for t in (c.xyz, c.density, c.scales, c.rotations))
dL = S.make_pixel_grad(512, 512).to(dev)


def step(i):
    v = views[i % 50]
    rs = GaussianRasterizationSettings(image_height=512, image_width=512, tanfovx=v.tanfovx, tanfovy=v.tanfovy, scale_modifier=1.0,
                                       viewmatrix=v.world_view_transform.to(dev), projmatrix=v.full_proj_transform.to(dev),
                                       campos=v.camera_center.to(dev), prefiltered=False, mode=v.mode, debug=False)
    m2d = torch.zeros(300000, 3, device=dev, requires_grad=True)
    img, _ = GaussianRasterizer(rs)(xyz, m2d, dens, sc, rot)
    img.backward(dL)


for i in range(60):   # no synchronisation in between: the stamps of the last step show the steady state
    step(i)
torch.cuda.synchronize()
L = _lib.lib()
tabs = {}
t0 = None
for u in ("order", "geom", "sort", "render"):
    try:
        f = getattr(L, "r2_debug_ts_" + u)
    except AttributeError:
        continue
    a = np.zeros(16 * 2048, dtype=np.uint64)
    f(a.ctypes.data_as(C.c_void_p))
    tabs[u] = a.reshape(16, 2048)
    nz = a[a > 0]
    if nz.size:
        t0 = nz.min() if t0 is None else min(t0, nz.min())
for u, t in tabs.items():
    for ph in range(16):
        v = t[ph][t[ph] > 0]
        if v.size:
            v = np.sort((v - t0).astype(np.float64) * 0.01)
            print("  TS %-6s %2d: n %4d  min %7.2f  med %7.2f  max %7.2f us" % (u, ph, v.size, v[0], v[v.size // 2], v[-1]))
