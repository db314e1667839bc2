"""Import-name shim: lets unmodified reference code (r2_gaussian/gaussian/render_query.py:14-19) run on
the MI355X kernels.  Same exports as SUB/xray_gaussian_rasterization_voxelization/__init__.py:1-2."""
from r2_gaussian_amd.rasterization import GaussianRasterizationSettings, GaussianRasterizer   # noqa: F401
from r2_gaussian_amd.voxelization import GaussianVoxelizationSettings, GaussianVoxelizer      # noqa: F401
from r2_gaussian_amd import _C                                                                # noqa: F401
