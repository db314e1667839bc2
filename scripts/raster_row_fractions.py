"""How much of the render kernels' pixel arithmetic could row-granular culling remove?  (round 5, CPU / oracle only.)
The kernels cull per 8x8 block by the bounding box of {alpha >= 1e-5}; a live (instance, block) item then evaluates all 8 rows.
For the headline scene: the share of those rows that (a) intersect the bounding box in y, (b) hold at least one pixel of the exact
region -- and, for scale, the share of pixels that pass.    python scripts/raster_row_fractions.py"""
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
    n_items = rows_box = rows_exact = pix = 0
    rng = np.random.default_rng(0)
    sel = rng.choice(len(ids), size=min(len(ids), 200000), replace=False)   # a sample of the instances
    g = ids[sel]
    px, py = m2[g, 0], m2[g, 1]
    for bx in (0, 8):
        for by in (0, 8):
            x0, y0 = tx[sel] * 16.0 + bx, ty[sel] * 16.0 + by
            live = (px - hx[g] <= x0 + 7) & (px + hx[g] >= x0) & (py - hy[g] <= y0 + 7) & (py + hy[g] >= y0)
            k = np.nonzero(live)[0]
            n_items += len(k)
            ys = y0[k, None] + np.arange(8)[None, :]
            rows_box += (np.abs(ys - py[k, None]) <= hy[g[k], None]).sum()
            xs = x0[k, None, None] + np.arange(8)[None, None, :]
            dx, dy = px[k, None, None] - xs, py[k, None, None] - ys[:, :, None]
            q = A[g[k], None, None] * dx * dx + 2 * B[g[k], None, None] * dx * dy + C[g[k], None, None] * dy * dy
            ok = q <= qmax[g[k], None, None]
            rows_exact += ok.any(axis=2).sum()
            pix += ok.sum()
    print("%s: live (instance, block) items per instance %.2f; of their rows: %.1f %% meet the box in y, %.1f %% hold a passing pixel; "
          "%.1f %% of their pixels pass" % (label, n_items / len(sel), 100.0 * rows_box / (8 * n_items), 100.0 * rows_exact / (8 * n_items),
                                            100.0 * pix / (64 * n_items)))


if __name__ == "__main__":
    views = S.make_views(50, (512, 512))
    stats(S.make_cloud(300000, seed=0), views[0], "synthetic 300k / 512^2 view 0")
    stats(S.make_cloud(300000, seed=0), views[17], "synthetic 300k / 512^2 view 17")
