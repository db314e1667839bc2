"""How many (tile, Gaussian) instances of the rasterizer's lists cannot contribute to their tile at all?  (VERDICT r3 #3b: measure
the rasterizer's share before building a render-only skip list.)  An instance exists because the Gaussian's 3-sigma SQUARE touches
the tile (the reference's contract, RAS/auxiliary.h:50-60); it is dead when the bounding box of {alpha >= 1e-5} -- the test the
render kernels already apply per 8x8 block -- misses the whole 16x16 tile, and "no pixel passes" when the exact region does.
CPU only (oracle).    python scripts/raster_dead_instances.py [small large]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O   # noqa: E402
from r2_gaussian_amd import scene as S   # noqa: E402


def stats(c, v, label):
    x, d, s, r = (t.numpy() for t in (c.xyz, c.density, c.scales, c.rotations))
    o = O.raster_forward(x, d, s, r, 1.0, None, v.world_view_transform.numpy(), v.full_proj_transform.numpy(), v.tanfovx, v.tanfovy,
                         v.image_height, v.image_width, v.mode, render=False)
    R = o["num_rendered"]
    gxn = o["grid"][0]
    rg = o["ranges"].astype(np.int64)
    tiles = np.repeat(np.arange(rg.shape[0]), rg[:, 1] - rg[:, 0])
    ids = o["point_list"].astype(np.int64)
    m2, con, mu = o["means2D"].astype(np.float64), o["conic_opacity"].astype(np.float64), o["mus"].astype(np.float64)
    A, B, C, w = con[:, 0], con[:, 1], con[:, 2], con[:, 3] * mu
    with np.errstate(divide="ignore", invalid="ignore"):
        qmax = 2.0 * (np.log(w) - np.log(1e-5))
        det = A * C - B * B
        hx = np.where(qmax > 0, np.sqrt(np.maximum(qmax, 0) * C / det), -np.inf)
        hy = np.where(qmax > 0, np.sqrt(np.maximum(qmax, 0) * A / det), -np.inf)
    tx, ty = tiles % gxn, tiles // gxn
    px, py = m2[ids, 0], m2[ids, 1]
    x0, y0 = tx * 16.0, ty * 16.0
    live_box = (px - hx[ids] <= x0 + 15) & (px + hx[ids] >= x0) & (py - hy[ids] <= y0 + 15) & (py + hy[ids] >= y0)
    dead_box = ~live_box
    idx = np.nonzero(live_box)[0]
    live_exact = np.zeros(R, bool)
    nblk = np.zeros(R, np.int64)
    for a in range(0, len(idx), 100000):
        k = idx[a:a + 100000]
        gxs = x0[k, None, None] + np.arange(16)[None, None, :]
        gys = y0[k, None, None] + np.arange(16)[None, :, None]
        dx, dy = px[k, None, None] - gxs, py[k, None, None] - gys
        q = A[ids[k], None, None] * dx * dx + 2 * B[ids[k], None, None] * dx * dy + C[ids[k], None, None] * dy * dy
        ok = q <= qmax[ids[k], None, None]
        live_exact[k] = ok.any(axis=(1, 2))
    # live 8x8 blocks per instance by the kernels' box test
    for bx in (0, 8):
        for by in (0, 8):
            nblk += ((px - hx[ids] <= x0 + bx + 7) & (px + hx[ids] >= x0 + bx) & (py - hy[ids] <= y0 + by + 7) & (py + hy[ids] >= y0 + by))
    print("%s: R %d  box-dead %.2f %%  no-pixel-passes %.2f %%  live 8x8 blocks per instance %.2f (of 4)" % (
        label, R, 100.0 * dead_box.mean(), 100.0 * (~live_exact).mean(), nblk.mean()))


if __name__ == "__main__":
    views = S.make_views(50, (512, 512))
    stats(S.make_cloud(300000, seed=0), views[0], "synthetic 300k / 512^2 view 0")
    from tests import trained_cloud as TC
    for name in sys.argv[1:] or ["small", "large"]:
        c, _info = TC.load(name, train=False)
        if c is not None:
            stats(c, views[0], "trained %s (P %d) view 0" % (name, c.xyz.shape[0]))
