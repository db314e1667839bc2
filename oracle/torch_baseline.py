"""Pure-PyTorch (float32, CPU) evaluation of the hot path's math: the CPU baseline that BASELINE.json's north_star and
SURVEY.md 8(d) name for configuration A (0_chest_cone-like, 5k Gaussians, 64 x 64 detector, 10 views, + a 64^3 volume query).

TEST INFRASTRUCTURE / REPORTED BASELINE, like everything under oracle/: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this file.  It is NOT the parity oracle (that is oracle/r2_oracle.c, pinned bit-exactly against the
reference's own kernels compiled on the CPU): torch's vectorised float32 ops round differently from the scalar kernels, so this
file is checked against the oracle at 1e-4 relative (tests/test_torch_baseline_cpu.py), not bit for bit.  What it keeps exactly is
the *algorithm*: tile-exact lists (a pixel only accumulates Gaussians whose tile rectangle contains its tile: truncating at the
3-sigma box is part of the reference's result, RAS/forward.cu:198-289 + RAS/auxiliary.h:50-60), the power > 0 and alpha cut-offs
(RAS/forward.cu:361-376, VOX/forward.cu:274-283), the cone / parallel Jacobians (RAS/forward.cu:77-156).  The backward is torch
autograd of that forward -- "the same math" as a user of PyTorch alone would write it; where the reference's hand-written
backward deviates from the true gradient on purpose (clamped-t Jacobian terms) autograd differs,
which is why this is a baseline and not an oracle.

Everything is vectorised over Gaussians; the render loops over tiles in Python (16 tiles at 64 x 64).
"""
import math

import torch


def cov3d(scales, scale_modifier, rotations):
    """Sigma = (S R)^T (S R), quaternion (r, x, y, z) not normalised (RAS/forward.cu:161-195).  Returns the 6 upper entries."""
    r, x, y, z = rotations.unbind(-1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    # (the kernels fill glm's column-major mat3 with this row-major listing, i.e. hold R^T, and form (S R^T)^T (S R^T)
    #  = R S^2 R^T: the standard covariance)
    M = R * (scale_modifier * scales)[:, None, :]
    S = M @ M.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1)


def _sym(c6):
    a, b, c, d, e, f = c6.unbind(-1)
    return torch.stack([a, b, c, b, d, e, c, e, f], -1).reshape(-1, 3, 3)


def raster_preprocess(means3D, opacities, scales, rotations, scale_modifier, viewmatrix, projmatrix, tanfovx, tanfovy, H, W, mode):
    """RAS/forward.cu:198-289, vectorised.  viewmatrix / projmatrix: the [4,4] tensors of the camera (row-vector convention:
    p_view = [p, 1] @ viewmatrix), i.e. what the kernels index column-major."""
    P = means3D.shape[0]
    ones = torch.ones(P, 1, dtype=means3D.dtype)
    ph = torch.cat([means3D, ones], 1)
    p_view = (ph @ viewmatrix)[:, :3]
    p_hom = ph @ projmatrix
    p_w = 1.0 / (p_hom[:, 3] + 0.0000001)
    projx, projy = p_hom[:, 0] * p_w, p_hom[:, 1] * p_w
    focal_y, focal_x = H / (2.0 * tanfovy), W / (2.0 * tanfovx)
    c3 = cov3d(scales, scale_modifier, rotations)
    tx, ty, tz = p_view.unbind(-1)
    zeros = torch.zeros_like(tx)
    if mode == 0:   # parallel beam (RAS/forward.cu:96-108; the y limit uses limx there, Q1)
        tx = tx.clamp(-1.3, 1.3)
        ty = ty.clamp(-1.3, 1.3)
        J = torch.stack([focal_x + zeros, zeros, zeros, zeros, focal_y + zeros, zeros, zeros, zeros, 1 + zeros], -1)
    else:           # cone beam (RAS/forward.cu:109-130)
        limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
        tx = (tx / tz).clamp(-limx, limx) * tz
        ty = (ty / tz).clamp(-limy, limy) * tz
        l = torch.sqrt(tx * tx + ty * ty + tz * tz)
        J = torch.stack([focal_x / tz, zeros, -(focal_x * tx) / (tz * tz),
                         zeros, focal_y / tz, -(focal_y * ty) / (tz * tz),
                         tx / l, ty / l, tz / l], -1)
    J = J.reshape(-1, 3, 3)                      # row-major: rows = the three output coordinates
    Wm = viewmatrix[:3, :3].transpose(0, 1)      # world -> view rotation (row-major, column vectors)
    T = J @ Wm                                   # [P,3,3]
    cov = T @ _sym(c3) @ T.transpose(1, 2)
    a, b, c = cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2]
    d, e, f = cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]
    diamond = a * d - b * b
    circ = a * d * f + 2 * b * c * e - a * e * e - f * b * b - d * c * c
    mu_sq = 2 * math.pi * circ / diamond
    mu = torch.where(mu_sq > 0, torch.sqrt(mu_sq.clamp_min(1e-38)), torch.zeros_like(mu_sq))
    det = diamond
    det_inv = 1.0 / det
    conic = torch.stack([d * det_inv, -b * det_inv, a * det_inv], -1)
    mid = 0.5 * (a + d)
    root = torch.sqrt((mid * mid - det).clamp_min(0.1))
    radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(mid + root, mid - root)))
    px = ((projx.double() + 1.0) * W - 1.0) * 0.5
    py = ((projy.double() + 1.0) * H - 1.0) * 0.5
    px, py = px.float(), py.float()
    gx, gy = (W + 15) // 16, (H + 15) // 16
    with torch.no_grad():
        rad = radius.detach()

        def cl(v, g):   # C-style (int) truncation, then clamp to [0, g]
            return v.trunc().clamp(0, g).to(torch.int64)
        x0, y0 = cl((px - rad) / 16, gx), cl((py - rad) / 16, gy)
        x1, y1 = cl((px + rad + 15) / 16, gx), cl((py + rad + 15) / 16, gy)
        vis = (p_view[:, 2] > 0.2) & (det != 0) & ((x1 - x0) * (y1 - y0) > 0)
    return dict(vis=vis, radii=torch.where(vis, rad, torch.zeros_like(rad)).to(torch.int32), px=px, py=py, conic=conic, mu=mu,
                rect=(x0, y0, x1, y1), depth=p_view[:, 2], grid=(gx, gy))


def rasterize(means3D, opacities, scales, rotations, scale_modifier, viewmatrix, projmatrix, tanfovx, tanfovy, H, W, mode):
    """[1,H,W] line integrals (RAS/forward.cu:294-395), differentiable through autograd."""
    g = raster_preprocess(means3D, opacities, scales, rotations, scale_modifier, viewmatrix, projmatrix, tanfovx, tanfovy, H, W, mode)
    gx, gy = g["grid"]
    x0, y0, x1, y1 = g["rect"]
    amp = opacities.reshape(-1) * g["mu"]
    rows = []
    ly, lx = torch.meshgrid(torch.arange(16), torch.arange(16), indexing="ij")
    for ty in range(gy):
        row = []
        for tx in range(gx):
            ids = torch.nonzero(g["vis"] & (x0 <= tx) & (tx < x1) & (y0 <= ty) & (ty < y1)).reshape(-1)
            pixx = (tx * 16 + lx).reshape(-1, 1).float()
            pixy = (ty * 16 + ly).reshape(-1, 1).float()
            dx = g["px"][ids][None, :] - pixx
            dy = g["py"][ids][None, :] - pixy
            co = g["conic"][ids]
            power = -0.5 * (co[:, 0] * dx * dx + co[:, 2] * dy * dy) - co[:, 1] * dx * dy
            alpha = amp[ids][None, :] * torch.exp(power)
            keep = (power <= 0) & (alpha >= 0.00001)
            row.append(torch.where(keep, alpha, torch.zeros_like(alpha)).sum(1).reshape(16, 16))
        rows.append(torch.cat(row, 1))
    img = torch.cat(rows, 0)[:H, :W]
    return img[None], g


def voxelize(means3D, opacities, scales, rotations, scale_modifier, nVoxel, sVoxel, center):
    """[nx,ny,nz] densities (VOX/forward.cu:58-178, 183-315), differentiable through autograd."""
    nx, ny, nz = (int(v) for v in nVoxel)
    sx, sy, sz = (float(v) for v in sVoxel)
    cx, cy, cz = (float(v) for v in center)
    dv = torch.tensor([sx / nx, sy / ny, sz / nz])
    c3 = cov3d(scales, scale_modifier, rotations)
    cov = _sym(c3) / (dv[:, None] * dv[None, :])
    inv = torch.linalg.inv(cov)
    pv = (means3D - torch.tensor([cx, cy, cz]) + torch.tensor([sx, sy, sz]) / 2) / dv
    n = torch.tensor([nx, ny, nz])
    g3 = (n + 7) // 8
    with torch.no_grad():
        rad = torch.ceil(3.0 * scales.max(1).values[:, None] / dv)     # raw scales, no modifier (Q5)
        inside = ((pv + rad >= 0) & (pv - rad <= n)).all(1)
        lo = ((pv - rad) / 8).trunc().clamp_min(0).minimum(g3).to(torch.int64)
        hi = ((pv + rad + 7) / 8).trunc().clamp_min(0).minimum(g3).to(torch.int64)
        vis = inside & ((hi - lo).prod(1) > 0) & (torch.linalg.det(cov) != 0)
    amp = opacities.reshape(-1)
    vol = torch.zeros(nx, ny, nz)
    l = torch.arange(8)
    lx, ly, lz = torch.meshgrid(l, l, l, indexing="ij")
    out = []
    gxn, gyn, gzn = (int(v) for v in g3)
    for tx in range(gxn):
        plane = []
        for ty in range(gyn):
            line = []
            for tz in range(gzn):
                t = torch.tensor([tx, ty, tz])
                ids = torch.nonzero(vis & (lo <= t).all(1) & (t < hi).all(1)).reshape(-1)
                f = torch.stack([tx * 8 + lx, ty * 8 + ly, tz * 8 + lz], -1).reshape(-1, 1, 3).float() + 0.5
                d = pv[ids][None, :, :] - f                                  # [512, n, 3]
                power = -0.5 * torch.einsum("vni,nij,vnj->vn", d, inv[ids], d)
                alpha = amp[ids][None, :] * torch.exp(power)
                keep = (power <= 0) & (alpha >= 0.000001)
                line.append(torch.where(keep, alpha, torch.zeros_like(alpha)).sum(1).reshape(8, 8, 8))
            plane.append(torch.cat(line, 2))
        out.append(torch.cat(plane, 1))
    vol = torch.cat(out, 0)[:nx, :ny, :nz]
    return vol, dict(vis=vis, radii=rad.to(torch.int32))


def config_a(n_gaussians=5000, detector=64, n_views=10, n_voxel=64, seed=0, backward=True):
    """Configuration A end to end (forward + autograd backward of every view, one volume query forward + backward) on the
    synthetic 0_chest_cone-like scene of r2_gaussian_amd.scene.  Returns (seconds, views, images, volume)."""
    import time

    from r2_gaussian_amd import scene as S
    cloud = S.make_cloud(n_gaussians, seed=seed)
    views = S.make_views(n_views, (detector, detector))
    params = [t.clone().requires_grad_(backward) for t in (cloud.xyz, cloud.density, cloud.scales, cloud.rotations)]
    t0 = time.perf_counter()
    images = []
    for v in views:
        img, _ = rasterize(*params, 1.0, v.world_view_transform, v.full_proj_transform, v.tanfovx, v.tanfovy,
                           v.image_height, v.image_width, v.mode)
        if backward:
            img.sum().backward()
        images.append(img.detach())
    sc = S.CONE_BEAM
    vol, _ = voxelize(*params, 1.0, [n_voxel] * 3, sc["sVoxel"], sc["offOrigin"])
    if backward:
        vol.sum().backward()
    return time.perf_counter() - t0, n_views, images, vol.detach()


def one_view(n_gaussians=300000, detector=512, n_views=50, view=0, seed=0, backward=True):
    """ONE view (forward + autograd backward) of a large workload -- SURVEY.md 8(d): "for B / C / E time one view and
    extrapolate".  Default = bench.py's headline workload (300k Gaussians, 512^2, view 0 of 50).  -> (seconds, num tiles)."""
    import time

    from r2_gaussian_amd import scene as S
    cloud = S.make_cloud(n_gaussians, seed=seed)
    v = S.make_views(n_views, (detector, detector))[view]
    params = [t.clone().requires_grad_(backward) for t in (cloud.xyz, cloud.density, cloud.scales, cloud.rotations)]
    t0 = time.perf_counter()
    img, _ = rasterize(*params, 1.0, v.world_view_transform, v.full_proj_transform, v.tanfovx, v.tanfovy, v.image_height,
                       v.image_width, v.mode)
    if backward:
        img.sum().backward()
    return time.perf_counter() - t0, ((detector + 15) // 16) ** 2
