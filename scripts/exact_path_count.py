"""How many visible Gaussians take the render kernels' exact path (item_tier != 0), as the preprocess counts them (host word DW_USER),
and what that means for the steps: a step of 64 entries runs the exact body too as soon as one of its lanes needs it.
   python scripts/exact_path_count.py"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from r2_gaussian_amd import scene as S
from tests import helpers as Hh

dev = torch.device("cuda:0")
cases = [("headline 300k/512", S.make_cloud(300000, seed=0), 512), ("B 50k/512", S.make_cloud(50000, seed=0), 512),
         ("E 1M/1024", S.make_cloud(1000000, seed=0), 1024)]
try:
    from tests import trained_cloud as TCl
    for name in ("small", "large"):
        c, info = TCl.load(name)
        cases.append(("trained " + name, c, 512))
except Exception as ex:   # noqa
    print("no trained clouds:", ex)
for name, c, hw in cases:
    v = S.make_views(50, (hw, hw))[7]
    Hh.hip_raster(c, v, dev)
    h = Hh.hip_raster(c, v, dev)
    nvis = int((h["radii"] > 0).sum())
    nex = int(h["host_words"][2])
    f = nex / max(nvis, 1)
    print("%-20s visible %8d  exact-path %7d = %.3f %%   P(step of 64 has one) = %.3f" % (name, nvis, nex, 100 * f, 1 - (1 - f) ** 64), flush=True)
