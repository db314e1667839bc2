"""Model / volume I/O compatible with the reference's files, so that models trained with the reference can
be loaded and evaluated on the MI355X voxelizer and vice versa:

* ``point_cloud/iteration_N/point_cloud.pickle`` -- dict(xyz, density, scale, rotation, scale_bound) of RAW (pre-activation)
  parameters as numpy arrays (GaussianModel.save_ply / load_ply, r2_gaussian/gaussian/gaussian_model.py:263-318);
* ``vol_gt.npy`` / ``vol_pred.npy``              (Scene.save, r2_gaussian/dataset/__init__.py:79-93; test.py:128-129);
* ``ckpt/chkpnt{it}.pth`` = ``torch.save((capture(), iteration))`` with the 10-tuple of GaussianModel.capture()
  (gaussian_model.py:79-110, train.py:185-190);
* the activations that turn the raw parameters into what the kernels consume (gaussian_model.py:38-64, 112-126);
* the full-volume evaluation of test.py:93-150: query at the scanner's nVoxel + 3D PSNR / SSIM (utils/image_utils.py:90-132).

Host logic only (pickle / numpy / torch); the volume query itself runs on the HIP voxelizer.
"""
import math
import os
import pickle

import numpy as np
import torch
import torch.nn.functional as F

CAPTURE_FIELDS = ("xyz", "scaling", "rotation", "density", "max_radii2D", "xyz_gradient_accum", "denom", "optimizer_state",
                  "spatial_lr_scale", "scale_bound")


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def save_point_cloud(path, xyz, density, scaling, rotation, scale_bound=None):
    """point_cloud.pickle (the reference "saves pickle rather than ply"): RAW parameters."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    out = {"xyz": _np(xyz), "density": _np(density), "scale": _np(scaling), "rotation": _np(rotation),
           "scale_bound": None if scale_bound is None else np.asarray(scale_bound)}
    with open(path, "wb") as f:
        pickle.dump(out, f, pickle.HIGHEST_PROTOCOL)


def load_point_cloud(path, device="cuda"):
    """-> dict(xyz, density, scaling, rotation: float32 tensors on `device`; scale_bound)."""
    with open(path, "rb") as f:
        d = pickle.load(f)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float, device=device)
    return {"xyz": t(d["xyz"]), "density": t(d["density"]), "scaling": t(d["scale"]), "rotation": t(d["rotation"]),
            "scale_bound": d.get("scale_bound")}


def activate(model):
    """raw -> (xyz[P,3], density[P,1] softplus, scaling[P,3] bounded sigmoid (or exp without a bound), rotation[P,4] unit)."""
    sb = model.get("scale_bound")
    if sb is not None:
        lo, hi = float(sb[0]), float(sb[1])
        scaling = torch.sigmoid(model["scaling"]) * (hi - lo) + lo
    else:
        scaling = torch.exp(model["scaling"])
    return model["xyz"], F.softplus(model["density"]), scaling, F.normalize(model["rotation"])


def save_volumes(directory, vol_gt, vol_pred):
    os.makedirs(directory, exist_ok=True)
    np.save(os.path.join(directory, "vol_gt.npy"), _np(vol_gt))
    np.save(os.path.join(directory, "vol_pred.npy"), _np(vol_pred))


def save_checkpoint(path, capture, iteration):
    """capture = the 10-tuple of GaussianModel.capture() (or a dict with CAPTURE_FIELDS)."""
    if isinstance(capture, dict):
        capture = tuple(capture[k] for k in CAPTURE_FIELDS)
    assert len(capture) == len(CAPTURE_FIELDS)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save((tuple(capture), int(iteration)), path)


def load_checkpoint(path, map_location=None):
    """-> (dict over CAPTURE_FIELDS, iteration)."""
    capture, iteration = torch.load(path, map_location=map_location, weights_only=False)
    assert len(capture) == len(CAPTURE_FIELDS), "not a GaussianModel.capture() checkpoint"
    return dict(zip(CAPTURE_FIELDS, capture)), int(iteration)


# ---------------------------------------------------------------------------------------------- evaluation (test.py:93-150)
def _ssim2d(a, b, window_size=11):
    """ssim of utils/loss_utils.py:57-104 on [1,1,H,W] slices."""
    g = torch.tensor([math.exp(-((x - window_size // 2) ** 2) / (2 * 1.5 ** 2)) for x in range(window_size)], dtype=a.dtype)
    g = (g / g.sum()).unsqueeze(1)
    w = (g @ g.t())[None, None].to(a.device)
    pad = window_size // 2
    mu1, mu2 = F.conv2d(a, w, padding=pad), F.conv2d(b, w, padding=pad)
    s1 = F.conv2d(a * a, w, padding=pad) - mu1 * mu1
    s2 = F.conv2d(b * b, w, padding=pad) - mu2 * mu2
    s12 = F.conv2d(a * b, w, padding=pad) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))).mean()


@torch.no_grad()
def metric_vol(vol_gt, vol_pred, metric="psnr", pixel_max=1.0):
    """metric_vol of utils/image_utils.py:90-132: 3D PSNR, or the mean over the three axes of the slice-wise 2D SSIM (slices
    whose ground truth is all zero count as 0 and are left out of the denominator)."""
    a, b = torch.as_tensor(vol_gt).float(), torch.as_tensor(vol_pred).float()
    if metric == "psnr":
        pm = float(a.max()) if pixel_max is None else pixel_max
        return float(10 * torch.log10(pm ** 2 / torch.mean((a - b) ** 2))), None
    assert metric == "ssim"
    per_axis = []
    for axis in range(3):
        total, count = 0.0, 0
        for i in range(a.shape[axis]):
            s1, s2 = a.select(axis, i), b.select(axis, i)
            if float(s1.max()) > 0:
                total += float(_ssim2d(s1[None, None], s2[None, None]))
                count += 1
        per_axis.append(total / max(count, 1))
    return float(np.mean(per_axis)), per_axis


@torch.no_grad()
def evaluate_volume(model, scanner_cfg, vol_gt=None, save_dir=None):
    """query() of render_query.py:27-77 at the scanner's full resolution on the HIP voxelizer, + test.py's 3D metrics.
    model: dict from load_point_cloud (raw parameters).  -> dict(vol, psnr_3d, ssim_3d, ...)."""
    from .voxelization import GaussianVoxelizationSettings, GaussianVoxelizer
    xyz, dens, scal, rot = activate(model)
    n, s, c = scanner_cfg["nVoxel"], scanner_cfg["sVoxel"], scanner_cfg["offOrigin"]
    vs = GaussianVoxelizationSettings(scale_modifier=1.0, nVoxel_x=int(n[0]), nVoxel_y=int(n[1]), nVoxel_z=int(n[2]),
                                      sVoxel_x=float(s[0]), sVoxel_y=float(s[1]), sVoxel_z=float(s[2]),
                                      center_x=float(c[0]), center_y=float(c[1]), center_z=float(c[2]),
                                      prefiltered=False, debug=False)
    vol, radii = GaussianVoxelizer(voxel_settings=vs)(means3D=xyz, opacities=dens, scales=scal, rotations=rot, cov3D_precomp=None)
    out = {"vol": vol, "radii": radii}
    if vol_gt is not None:
        gt = torch.as_tensor(vol_gt).to(vol.device)
        out["psnr_3d"] = metric_vol(gt, vol, "psnr")[0]
        out["ssim_3d"], (out["ssim_3d_x"], out["ssim_3d_y"], out["ssim_3d_z"]) = metric_vol(gt.cpu(), vol.cpu(), "ssim")
        if save_dir:
            save_volumes(save_dir, gt, vol)
    return out
