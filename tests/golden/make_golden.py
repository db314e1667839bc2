#!/usr/bin/env python
"""Generates tests/golden/*.npz -- the committed golden vectors that pin the CPU oracle.

Run in the build container only (it needs /root/reference):   python tests/golden/make_golden.py

Two sources, both the REFERENCE ITSELF executed here:

1. ``oracle/_ref`` -- the reference's own CUDA kernels (SUB/cuda_rasterizer/*.cu, SUB/cuda_voxelizer/*.cu) compiled
   for the host CPU from where they lie (``make -C oracle ref``; see oracle/Makefile for the shim boundary).  Full
   forward states (radii, tile counts, (tile|depth) keys, sorted point lists, ranges, n_contrib, images / volumes) and
   all backward outputs on small seeded scenes -> ``raster_*.npz``, ``voxel_*.npz``.
2. the reference's *Python* statements, imported from /root/reference: camera matrices
   (``angle2pose`` dataset_readers.py:156-191, ``getWorld2View2`` / ``getProjectionMatrix`` graphics_utils.py:81-142,
   composed as ``Camera.__init__`` does, cameras.py:66-84) and the covariance Sigma = R S^2 R^T
   (``build_scaling_rotation`` / ``strip_symmetric`` gaussian_utils.py:49-100, composed as
   gaussian_model.py:38-42) -> ``camera.npz``, ``cov3d.npz``; 3D PSNR (``metric_vol`` image_utils.py:90-104) -> ``psnr.npz``.

The inputs are stored next to the outputs, so the tests need neither the reference tree nor this script.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import ref as Rf                     # noqa: E402
from r2_gaussian_amd import scene as S           # noqa: E402


def _load_ref_module(relpath, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print("wrote %-34s %7.1f KiB" % (name, os.path.getsize(path) / 1024.0))


def flat(prefix, d):
    return {prefix + k: np.asarray(v) for k, v in d.items() if isinstance(v, np.ndarray)}


def raster_case(name, P, H, W, scanner, angle, seed, scale_mult=1.0, scale_modifier=1.0, precomp=False, tweak=None):
    c = S.make_cloud(P, seed=seed, scanner=scanner, scale_mult=scale_mult)
    v = S.make_view(angle, (H, W), scanner)
    xyz, rho, sc, q = (t.numpy().copy() for t in (c.xyz, c.density, c.scales, c.rotations))
    if tweak:
        tweak(xyz, rho, sc, q)
    vm, pm = v.world_view_transform.numpy(), v.full_proj_transform.numpy()
    cov = None
    if precomp:   # cov3D_precomp path: feed the reference its own cov3D back, scales/rotations absent
        cov = Rf.raster_forward(xyz, rho, sc, q, scale_modifier, None, vm, pm, v.tanfovx, v.tanfovy, H, W, v.mode)["cov3D"]
        sc_in, q_in = None, None
    else:
        sc_in, q_in = sc, q
    st = Rf.raster_forward(xyz, rho, sc_in, q_in, scale_modifier, cov, vm, pm, v.tanfovx, v.tanfovy, H, W, v.mode)
    dL = S.make_pixel_grad(H, W, seed=seed + 100).numpy()
    g = Rf.raster_backward(st, xyz, sc_in, q_in, scale_modifier, cov, vm, pm, v.tanfovx, v.tanfovy, dL)
    ins = dict(in_means3D=xyz, in_opacities=rho, in_scales=sc, in_rotations=q, in_viewmatrix=vm, in_projmatrix=pm,
               in_dL_dcolor=dL, in_meta=np.array([H, W, v.mode, int(precomp)], np.int32),
               in_params=np.array([v.tanfovx, v.tanfovy, scale_modifier], np.float64))
    if cov is not None:
        ins["in_cov3D_precomp"] = cov
    save(name, num_rendered=np.array(st["num_rendered"]), **ins, **flat("fw_", st), **flat("bw_", g))


def voxel_case(name, P, nVoxel, sVoxel, center, seed, scale_mult=1.0, scale_modifier=1.0):
    c = S.make_cloud(P, seed=seed, scale_mult=scale_mult)
    xyz, rho, sc, q = (t.numpy().copy() for t in (c.xyz, c.density, c.scales, c.rotations))
    st = Rf.voxel_forward(xyz, rho, sc, q, scale_modifier, None, nVoxel, sVoxel, center)
    g_ = torch.Generator().manual_seed(seed + 200)
    n = int(np.prod(nVoxel))
    dL = ((torch.rand(*nVoxel, generator=g_) * 2 - 1) / n).numpy()
    g = Rf.voxel_backward(st, xyz, sc, q, scale_modifier, None, dL)
    save(name, num_rendered=np.array(st["num_rendered"]), in_means3D=xyz, in_opacities=rho, in_scales=sc, in_rotations=q,
         in_dL_dvol=dL, in_nVoxel=np.array(nVoxel, np.int32), in_sVoxel=np.array(sVoxel, np.float64),
         in_center=np.array(center, np.float64), in_params=np.array([scale_modifier], np.float64),
         **flat("fw_", st), **flat("bw_", g))


def python_goldens():
    sys.modules.setdefault("plyfile", types.SimpleNamespace(PlyData=None, PlyElement=None))
    gu = _load_ref_module("r2_gaussian/utils/graphics_utils.py", "_ref_graphics_utils")
    pkg = types.ModuleType("r2_gaussian"); pkg.__path__ = []
    utils = types.ModuleType("r2_gaussian.utils"); utils.__path__ = []
    sys.modules.update({"r2_gaussian": pkg, "r2_gaussian.utils": utils, "r2_gaussian.utils.graphics_utils": gu})
    dr = _load_ref_module("r2_gaussian/dataset/dataset_readers.py", "_ref_dataset_readers")
    ga = _load_ref_module("r2_gaussian/utils/gaussian_utils.py", "_ref_gaussian_utils")
    iu_src = open(os.path.join(REF, "r2_gaussian/utils/image_utils.py")).read()

    # ---- camera matrices, composed exactly as Camera.__init__ (cameras.py:66-84) minus the .cuda() calls
    out = {}
    angles = np.array([0.0, 0.7, 2.5, 4.0, 5.9])
    for mode_name, mode in (("parallel", 0), ("cone", 1)):
        cfg = dict(S.CONE_BEAM, mode=mode_name)
        wv, fp, cc, tf = [], [], [], []
        for a in angles:
            c2w = dr.angle2pose(cfg["DSO"], a)
            w2c = np.linalg.inv(c2w)
            R = np.transpose(w2c[:3, :3])
            T = w2c[:3, 3]
            FovX = np.arctan2(cfg["sDetector"][1] / 2, cfg["DSD"]) * 2
            FovY = np.arctan2(cfg["sDetector"][0] / 2, cfg["DSD"]) * 2
            wvt = torch.tensor(gu.getWorld2View2(R, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
            proj = gu.getProjectionMatrix(fovX=FovX, fovY=FovY, mode=mode, scanner_cfg=cfg).transpose(0, 1)
            full = wvt.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0)
            wv.append(wvt.numpy()); fp.append(full.numpy()); cc.append(wvt.inverse()[3, :3].numpy())
            # render() settings: tanfov = 1 for parallel beam (render_query.py:103-111)
            tf.append([1.0, 1.0] if mode == 0 else [np.tan(FovX * 0.5), np.tan(FovY * 0.5)])
        out.update({mode_name + "_world_view": np.stack(wv), mode_name + "_full_proj": np.stack(fp),
                    mode_name + "_center": np.stack(cc), mode_name + "_tanfov": np.array(tf)})
    save("camera.npz", angles=angles, **out)

    # ---- covariance: the reference allocates on device="cuda"; run the same code with the allocation on the CPU
    real_zeros = torch.zeros
    torch.zeros = lambda *a, **k: real_zeros(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
    try:
        c = S.make_cloud(64, seed=11)
        mod = 1.3
        L = ga.build_scaling_rotation(mod * c.scales, c.rotations)
        cov = ga.strip_symmetric(L @ L.transpose(1, 2))
    finally:
        torch.zeros = real_zeros
    save("cov3d.npz", scales=c.scales.numpy(), rotations=c.rotations.numpy(), scale_modifier=np.array(mod), cov3D=cov.numpy())

    # ---- 3D PSNR
    ns = {"torch": torch, "np": np}
    start = iu_src.index("@torch.no_grad()\ndef metric_vol")
    end = iu_src.index("elif metric == \"ssim\"", start)
    exec(iu_src[start:end].rstrip(), ns)   # the psnr branch of metric_vol only (ssim needs the rest of the file)
    g = torch.Generator().manual_seed(5)
    a = torch.rand(12, 10, 14, generator=g)
    b = a + 0.05 * torch.randn(12, 10, 14, generator=g)
    save("psnr.npz", vol_gt=a.numpy(), vol_pred=b.numpy(), psnr=np.array(ns["metric_vol"](a, b, "psnr")[0]))


def main():
    assert os.path.isdir(REF), "needs the reference tree"
    Rf.build()

    def degenerate(xyz, rho, sc, q):
        # edge cases of the reference's tests-that-do-not-exist: behind-the-source (near cull, z_view <= 0.2),
        # far outside the detector, needle-thin and huge Gaussians, exact duplicates (depth ties -> stable order)
        xyz[0] = [5.2, 0.0, 0.0]; xyz[1] = [4.9, 0.0, 0.1]          # beyond / right at the source plane
        xyz[2] = [0.0, 3.0, 3.0]                                      # projects off the detector
        sc[3] = [0.0005, 0.0005, 0.4]; sc[4] = [0.9, 0.9, 0.9]
        xyz[6] = xyz[5]; sc[6] = sc[5]; q[6] = q[5]                   # duplicate -> identical depth keys
        xyz[8] = xyz[7]
        rho[9] = 0.0

    raster_case("raster_cone_300_48x40.npz", 300, 48, 40, S.CONE_BEAM, 0.6, seed=3, scale_mult=2.5, tweak=degenerate)
    raster_case("raster_parallel_250_33x50.npz", 250, 33, 50, S.PARALLEL_BEAM, 2.2, seed=4, scale_mult=2.0)
    raster_case("raster_cone_precomp_200_32x32.npz", 200, 32, 32, S.CONE_BEAM, 4.1, seed=5, scale_mult=2.0,
                scale_modifier=1.3, precomp=True)
    raster_case("raster_cone_mod_200_64x64.npz", 200, 64, 64, S.CONE_BEAM, 1.3, seed=6, scale_mult=3.0, scale_modifier=0.7)
    voxel_case("voxel_200_20x16x24.npz", 200, (20, 16, 24), (2.0, 1.6, 2.4), (0.0, 0.0, 0.0), seed=7, scale_mult=2.0)
    voxel_case("voxel_tv_150_16x16x16.npz", 150, (16, 16, 16), (0.25, 0.25, 0.25), (0.2, -0.1, 0.3), seed=8, scale_mult=1.0,
               scale_modifier=1.0)
    python_goldens()


if __name__ == "__main__":
    main()
