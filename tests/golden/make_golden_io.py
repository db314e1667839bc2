"""tests/golden/model_io/: files written by the REFERENCE's own Python -- GaussianModel.save_ply (the point_cloud.pickle format,
r2_gaussian/gaussian/gaussian_model.py:263-281) and metric_vol(..., "ssim") (utils/image_utils.py:105-132) -- so that
r2_gaussian_amd/model_io.py is pinned against them (tests/test_model_io_cpu.py).  Needs /root/reference; missing
third-party modules the import chain touches but these functions never use (plyfile, simple_knn) are stubbed.

    python tests/golden/make_golden_io.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
OUT = os.path.join(HERE, "model_io")

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for name, attrs in (("plyfile", ("PlyData", "PlyElement")), ("simple_knn", ()), ("simple_knn._C", ("distCUDA2",))):
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, None)
        sys.modules[name] = m
    sys.path.insert(0, REF)
    from r2_gaussian.gaussian.gaussian_model import GaussianModel
    from r2_gaussian.utils.image_utils import metric_vol
    g = torch.Generator().manual_seed(11)
    P = 257
    bound = np.array([0.001, 1.0])
    gm = GaussianModel(bound)
    gm._xyz = torch.randn(P, 3, generator=g)
    gm._density = torch.randn(P, 1, generator=g)
    gm._scaling = torch.randn(P, 3, generator=g)
    gm._rotation = torch.randn(P, 4, generator=g)
    gm.save_ply(os.path.join(OUT, "point_cloud.pickle"))
    # activations evaluated by the reference's own properties
    np.savez_compressed(os.path.join(OUT, "activated.npz"), density=gm.get_density.numpy(), scaling=gm.get_scaling.numpy(),
                        rotation=gm.get_rotation.numpy())
    a = torch.rand(12, 10, 14, generator=g)
    a[3] = 0.0   # an all-zero ground-truth slice: counts as 0 and is left out of the denominator
    b = (a + 0.05 * torch.randn(12, 10, 14, generator=g)).clamp_min(0)
    ssim, per_axis = metric_vol(a, b, "ssim")
    psnr, _ = metric_vol(a, b, "psnr")
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), vol_gt=a.numpy(), vol_pred=b.numpy(), ssim=np.array(ssim),
                        ssim_axes=np.array(per_axis), psnr=np.array(psnr))
    print("wrote", sorted(os.listdir(OUT)))
