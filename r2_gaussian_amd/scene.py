"""Synthetic cone-beam scene + camera conventions (host logic, CPU, numpy/torch).

The reference's datasets (``0_chest_cone``, ``pine``) and TIGRE are not available offline, so the
benchmark and the parity tests use a seeded synthetic set with the reference scanner geometry
(``data_generator/synthetic_dataset/scanner/cone_beam.yml``: cone, DSD 7, DSO 5, sDetector 4x4,
sVoxel 2^3 -> scene_scale 1).  The camera matrices are re-derived here exactly the way the reference
builds them, because their memory layout *is* the kernel input contract:

* ``angle2pose``           r2_gaussian/dataset/dataset_readers.py:156-191
* ``R, T`` from ``c2w``    r2_gaussian/dataset/dataset_readers.py:119-127
* ``getWorld2View2``       r2_gaussian/utils/graphics_utils.py:81-92
* ``getProjectionMatrix``  r2_gaussian/utils/graphics_utils.py:95-142
* ``Camera`` transforms    r2_gaussian/dataset/cameras.py:66-84
* ``FovX/FovY``            r2_gaussian/dataset/dataset_readers.py:131-132
* ``render()`` settings    r2_gaussian/gaussian/render_query.py:102-125

tests/golden/ pins these against vectors produced by importing the reference's own Python.
"""
import math
from typing import NamedTuple

import numpy as np
import torch

CONE_BEAM = dict(mode="cone", DSD=7.0, DSO=5.0, sDetector=[4.0, 4.0], nVoxel=[256, 256, 256],
                 sVoxel=[2.0, 2.0, 2.0], offOrigin=[0.0, 0.0, 0.0], offDetector=[0.0, 0.0])
PARALLEL_BEAM = dict(CONE_BEAM, mode="parallel")


def angle2pose(DSO, angle):
    """c2w for a source at ``angle`` on the circle of radius DSO (three fixed-axis rotations)."""
    phi1 = -np.pi / 2
    R1 = np.array([[1.0, 0.0, 0.0], [0.0, np.cos(phi1), -np.sin(phi1)], [0.0, np.sin(phi1), np.cos(phi1)]])
    phi2 = np.pi / 2
    R2 = np.array([[np.cos(phi2), -np.sin(phi2), 0.0], [np.sin(phi2), np.cos(phi2), 0.0], [0.0, 0.0, 1.0]])
    R3 = np.array([[np.cos(angle), -np.sin(angle), 0.0], [np.sin(angle), np.cos(angle), 0.0], [0.0, 0.0, 1.0]])
    rot = np.dot(np.dot(R3, R2), R1)
    trans = np.array([DSO * np.cos(angle), DSO * np.sin(angle), 0])
    T = np.eye(4)
    T[:3, :3] = rot
    T[:3, 3] = trans
    return T


def world2view(R, t, translate=np.array([0.0, 0.0, 0.0]), scale=1.0):
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    cam_center = (C2W[:3, 3] + translate) * scale
    C2W[:3, 3] = cam_center
    return np.float32(np.linalg.inv(C2W))


def projection_matrix(fovX, fovY, mode):
    if mode == 0:
        return torch.eye(4)
    znear, zfar = 0.01, 100.0
    top = math.tan(fovY / 2) * znear
    right = math.tan(fovX / 2) * znear
    bottom, left = -top, -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


class View(NamedTuple):
    """What ``render()`` feeds the rasterizer for one projection angle."""
    angle: float
    mode: int
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    world_view_transform: torch.Tensor  # [4,4] f32, = W2C^T stored row-major
    full_proj_transform: torch.Tensor   # [4,4] f32
    camera_center: torch.Tensor         # [3]


def make_view(angle, nDetector=(512, 512), scanner=CONE_BEAM) -> View:
    mode = {"parallel": 0, "cone": 1}[scanner["mode"]]
    scale = 2.0 / max(scanner["sVoxel"])  # scene_scale, dataset_readers.py:62-76
    DSO, DSD = scanner["DSO"] * scale, scanner["DSD"] * scale
    sDet = [s * scale for s in scanner["sDetector"]]
    c2w = angle2pose(DSO, angle)
    w2c = np.linalg.inv(c2w)
    R = np.transpose(w2c[:3, :3])
    T = w2c[:3, 3]
    FovX = np.arctan2(sDet[1] / 2, DSD) * 2
    FovY = np.arctan2(sDet[0] / 2, DSD) * 2
    wvt = torch.tensor(world2view(R, T)).transpose(0, 1).contiguous()
    proj = projection_matrix(FovX, FovY, mode).transpose(0, 1)
    full = wvt.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0).contiguous()
    center = wvt.inverse()[3, :3].contiguous()
    if mode == 0:
        tfx = tfy = 1.0
    else:
        tfx, tfy = math.tan(FovX * 0.5), math.tan(FovY * 0.5)
    return View(float(angle), mode, int(nDetector[0]), int(nDetector[1]), tfx, tfy, wvt, full, center)


def make_views(n_views, nDetector=(512, 512), scanner=CONE_BEAM):
    """theta = linspace(0, 2pi, N+1)[:-1] (data_generator/synthetic_dataset/generate_data.py:47-50)."""
    angles = np.linspace(0.0, 2.0 * np.pi, n_views + 1)[:-1]
    return [make_view(a, nDetector, scanner) for a in angles]


class Cloud(NamedTuple):
    xyz: torch.Tensor       # [P,3]
    scales: torch.Tensor    # [P,3] activated
    rotations: torch.Tensor  # [P,4] unit quaternions (r,x,y,z)
    density: torch.Tensor   # [P,1] activated


def make_cloud(P, seed=0, scanner=CONE_BEAM, scale_mult=1.0) -> Cloud:
    """Seeded synthetic Gaussian cloud: 95 % uniform in an ellipsoid with semi-axes
    (0.8, 0.6, 0.8), 5 % uniform in [-1,1]^3; per-axis log-uniform scales around
    0.6*(V/P)^(1/3) clipped to the reference scale bound [0.0005, 0.5]*max(sVoxel)
    (arguments/__init__.py:27-28, train.py:59-61); random unit quaternions; density U(0.01, 0.3)."""
    g = torch.Generator().manual_seed(seed)
    n_box = P // 20
    n_ell = P - n_box
    d = torch.randn(n_ell, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True).clamp_min(1e-12)
    r = torch.rand(n_ell, 1, generator=g) ** (1.0 / 3.0)
    ell = d * r * torch.tensor([0.8, 0.6, 0.8])
    box = torch.rand(n_box, 3, generator=g) * 2.0 - 1.0
    xyz = torch.cat([ell, box], 0)
    xyz = xyz[torch.randperm(P, generator=g)].contiguous()
    V = 4.0 / 3.0 * math.pi * 0.8 * 0.6 * 0.8
    s0 = 0.6 * (V / max(P, 1)) ** (1.0 / 3.0) * scale_mult
    scales = s0 * torch.exp((torch.rand(P, 3, generator=g) * 2.0 - 1.0) * 0.7)
    vol2world = 2.0 / max(scanner["sVoxel"]) * max(scanner["sVoxel"])
    scales = scales.clamp(0.0005 * vol2world, 0.5 * vol2world)
    q = torch.randn(P, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True).clamp_min(1e-12)
    density = 0.01 + 0.29 * torch.rand(P, 1, generator=g)
    return Cloud(xyz.float(), scales.float().contiguous(), q.float().contiguous(), density.float())


def make_pixel_grad(H, W, seed=1):
    """Upstream gradient dL/dpix = U(-1,1)/(H*W)."""
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(1, H, W, generator=g) * 2.0 - 1.0) / float(H * W)).float()


def psnr3d(vol_gt, vol_pred, pixel_max=1.0):
    """3D PSNR as defined by metric_vol(..., 'psnr') (r2_gaussian/utils/image_utils.py:90-104)."""
    mse = torch.mean((vol_gt.double() - vol_pred.double()) ** 2)
    return float(10.0 * torch.log10(pixel_max ** 2 / mse))
