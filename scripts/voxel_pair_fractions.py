"""How much of the voxelizer's evaluated work is live?  (measure the surviving pair fraction BEFORE building a
4^3-subtile culling stage.)  For the 256^3 query of the benchmark cloud, per (Gaussian, tile) instance of the reference's lists
(3-sigma cube of the LARGEST scale, VOX/forward.cu:58-178):
  evaluated_now   voxels of the x-slabs (1x8x8) whose box touches the bounding box of {alpha >= 1e-6}: what the lane-per-entry
                  kernel walks (csrc/voxel_render.hip slab_live)
  subtile_4       voxels of the 4x4x4 subtiles that touch that bounding box
  slab_rows       voxels of the (x-slab, y-row) 1x1x8 rows that touch it (the finest culling a z-recurrence allows)
  live            voxels that actually pass the cut-off (alpha >= 1e-6 and power <= 0)
all as fractions of instances x 512.  CPU only (numpy), a sample of the Gaussians.

    python scripts/voxel_pair_fractions.py [n_sample]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from r2_gaussian_amd import scene as S  # noqa: E402


def main():
    n_sample = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    P, n = 300000, 256
    c = S.make_cloud(P, seed=0)
    sc = S.CONE_BEAM
    st = O.voxel_forward(c.xyz.numpy(), c.density.numpy(), c.scales.numpy(), c.rotations.numpy(), 1.0, None, [n] * 3, sc["sVoxel"],
                         sc["offOrigin"], render=False)
    vis = np.nonzero(st["radii_x"] > 0)[0]
    rng = np.random.default_rng(0)
    ids = rng.choice(vis, size=min(n_sample, len(vis)), replace=False)
    pv = st["means3D_norm"][ids].astype(np.float64)
    co = st["conic_opacity"][ids].astype(np.float64)
    rad = np.stack([st["radii_x"][ids], st["radii_y"][ids], st["radii_z"][ids]], 1).astype(np.float64)
    g = n // 8
    tot = dict(instances=0, evaluated_now=0, slab_cond=0, slab_exact=0, subtile_4=0, slab_rows=0, live=0, tiles_dead=0)
    for k in range(len(ids)):
        a, b, cc, d, e, f, op = co[k]
        C = np.array([[a, b, cc], [b, d, e], [cc, e, f]])
        if op <= 0:
            continue
        qmax = 2.0 * np.log(op / 1e-6)
        if qmax <= 0:
            continue
        cov = np.linalg.inv(C)
        h = np.sqrt(qmax * np.diag(cov)) * 1.004 + 0.05
        lo = np.clip(((pv[k] - rad[k]) / 8).astype(int), 0, g)
        hi = np.clip(((pv[k] + rad[k] + 7) / 8).astype(int), 0, g)
        # voxel centres covered by the cube, the exact test on all of them at once
        x = np.arange(lo[0] * 8, hi[0] * 8) + 0.5
        y = np.arange(lo[1] * 8, hi[1] * 8) + 0.5
        z = np.arange(lo[2] * 8, hi[2] * 8) + 0.5
        if len(x) == 0 or len(y) == 0 or len(z) == 0:
            continue
        dx, dy, dz = pv[k, 0] - x, pv[k, 1] - y, pv[k, 2] - z
        q = (a * dx[:, None, None] ** 2 + d * dy[None, :, None] ** 2 + f * dz[None, None, :] ** 2
             + 2 * b * dx[:, None, None] * dy[None, :, None] + 2 * cc * dx[:, None, None] * dz[None, None, :]
             + 2 * e * dy[None, :, None] * dz[None, None, :])
        live = (q <= qmax) & (q >= 0)
        # interval tests of the bounding box against cells of size s along each axis (a cell [c0, c0 + s) holds voxel centres
        # c0 + 0.5 .. c0 + s - 0.5: it touches the box iff p - h <= c0 + s - 0.5 and p + h >= c0 + 0.5)
        def touch(p, hh, c0, s):
            return (p - hh <= c0 + s - 0.5) & (p + hh >= c0 + 0.5)
        tx8, ty8, tz8 = (touch(pv[k, i], h[i], np.arange(lo[i], hi[i]) * 8.0, 8.0) for i in range(3))
        tx1 = touch(pv[k, 0], h[0], np.arange(lo[0] * 8, hi[0] * 8).astype(float), 1.0)
        ty1 = touch(pv[k, 1], h[1], np.arange(lo[1] * 8, hi[1] * 8).astype(float), 1.0)
        tx4, ty4, tz4 = (touch(pv[k, i], h[i], np.arange(lo[i] * 2, hi[i] * 2) * 4.0, 4.0) for i in range(3))
        ninst = (hi - lo).prod()
        tot["instances"] += ninst
        tot["live"] += live.sum()
        # x-slab (1 x 8 x 8): slab x touches, and the tile's y / z ranges touch
        tot["evaluated_now"] += tx1.sum() * ty8.sum() * tz8.sum() * 64
        tot["slab_rows"] += tx1.sum() * ty1.sum() * tz8.sum() * 8
        tot["subtile_4"] += tx4.sum() * ty4.sum() * tz4.sum() * 64
        tot["tiles_dead"] += ninst - tx8.sum() * ty8.sum() * tz8.sum()
        # slab_cond (round 4): the cross-section of the cut-off ellipsoid at the slab's x is an ellipse centred at
        # p_yz - k * dx with the half-extents hc * sqrt(1 - (dx / hx)^2); its bounding box against the tile's y / z range
        M = np.array([[d, e], [e, f]])
        Minv = np.linalg.inv(M)
        kyz = -Minv @ np.array([b, cc])            # conditional centre of (dy, dz) per unit dx
        hc = np.sqrt(qmax * np.diag(Minv)) * 1.004 + 0.05
        t = np.sqrt(np.clip(1.0 - (dx / h[0]) ** 2, 0.0, None))       # per slab x
        cy = pv[k, 1] - kyz[0] * dx
        cz = pv[k, 2] - kyz[1] * dx
        y8 = np.arange(lo[1], hi[1]) * 8.0
        z8 = np.arange(lo[2], hi[2]) * 8.0
        oky = (cy[:, None] - hc[0] * t[:, None] <= y8[None, :] + 7.5) & (cy[:, None] + hc[0] * t[:, None] >= y8[None, :] + 0.5)
        okz = (cz[:, None] - hc[1] * t[:, None] <= z8[None, :] + 7.5) & (cz[:, None] + hc[1] * t[:, None] >= z8[None, :] + 0.5)
        cond = tx1[:, None, None] & oky[:, :, None] & okz[:, None, :]
        tot["slab_cond"] += cond.sum() * 64
        lv = live.reshape(len(x), hi[1] - lo[1], 8, hi[2] - lo[2], 8).any(axis=(2, 4))
        assert not (lv & ~cond).any(), "the conditional test dropped a slab that holds a live voxel"
        tot["slab_exact"] += lv.sum() * 64
    v = tot["instances"] * 512.0
    print("sampled Gaussians %d, instances %d (%.1f per Gaussian)" % (len(ids), tot["instances"], tot["instances"] / len(ids)))
    print("instances whose tile the alpha >= 1e-6 box does not touch at all: %.3f" % (tot["tiles_dead"] / tot["instances"]))
    for kx in ("evaluated_now", "slab_cond", "slab_exact", "subtile_4", "slab_rows", "live"):
        print("%-14s %.4f of instances x 512 voxels" % (kx, tot[kx] / v))
    print("slab_cond / evaluated_now = %.3f   slab_exact / evaluated_now = %.3f" % (
        tot["slab_cond"] / tot["evaluated_now"], tot["slab_exact"] / tot["evaluated_now"]))
    print("live / evaluated_now = %.3f   subtile_4 / evaluated_now = %.3f   slab_rows / evaluated_now = %.3f" % (
        tot["live"] / tot["evaluated_now"], tot["subtile_4"] / tot["evaluated_now"], tot["slab_rows"] / tot["evaluated_now"]))


if __name__ == "__main__":
    main()
