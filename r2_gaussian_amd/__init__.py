"""r2_gaussian_amd -- MI355X-native (gfx950) hot path of R2-Gaussian: differentiable X-ray rasterizer,
3D voxelizer and simple-knn behind the reference's own Python surface.

    from r2_gaussian_amd import (GaussianRasterizationSettings, GaussianRasterizer,
                                 GaussianVoxelizationSettings, GaussianVoxelizer, distCUDA2)

Drop-in import names for unmodified reference code live in the top-level shim packages
``xray_gaussian_rasterization_voxelization`` and ``simple_knn``.
"""
from .rasterization import GaussianRasterizationSettings, GaussianRasterizer, GaussianRasterizerBatch   # noqa: F401
from .voxelization import GaussianVoxelizationSettings, GaussianVoxelizer      # noqa: F401
from ._C import distCUDA2                                                      # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "GaussianVoxelizationSettings",
           "GaussianVoxelizer", "distCUDA2"]
