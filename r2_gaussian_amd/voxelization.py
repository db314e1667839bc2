"""Drop-in for ``xray_gaussian_rasterization_voxelization.voxelization``
(SUB/xray_gaussian_rasterization_voxelization/voxelization.py): ``GaussianVoxelizationSettings``,
``GaussianVoxelizer`` and the ``_VoxelizeGaussians`` autograd function with the reference's argument
order, return values (``(vol[nx,ny,nz], (radii_x, radii_y, radii_z))``) and gradient tuple
``(means3D, opacities, scales, rotations, cov3Ds_precomp, None)`` (PY/voxelization.py:216-225).
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C
from .rasterization import _guarded


class GaussianVoxelizationSettings(NamedTuple):
    scale_modifier: float
    nVoxel_x: int
    nVoxel_y: int
    nVoxel_z: int
    sVoxel_x: float
    sVoxel_y: float
    sVoxel_z: float
    center_x: float
    center_y: float
    center_z: float
    prefiltered: bool
    debug: bool


class GaussianVoxelizationSlabSettings(NamedTuple):
    """New (not in the reference): the settings of the FULL volume plus a range of 8-voxel tile layers along x.  A voxelizer
    built on them returns the [x1 - x0, ny, nz] block of the full volume -- every rank of the sharded query evaluates the full
    grid's arithmetic and renders only its layers, so the concatenated blocks are bit-identical to the unsharded volume
    (``dist.slab_settings`` builds these; C ABI: r2_voxel_forward_slab)."""
    scale_modifier: float
    nVoxel_x: int
    nVoxel_y: int
    nVoxel_z: int
    sVoxel_x: float
    sVoxel_y: float
    sVoxel_z: float
    center_x: float
    center_y: float
    center_z: float
    prefiltered: bool
    debug: bool
    tile_x0: int
    tile_x1: int


def _slab_of(vs):
    """(tile_x0, tile_x1) of slab settings, None for the reference's settings."""
    t0 = getattr(vs, "tile_x0", None)
    return None if t0 is None else (int(t0), int(vs.tile_x1))


class _VoxelizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, opacities, scales, rotations, cov3Ds_precomp, voxel_settings):
        vs = voxel_settings
        args = (means3D, opacities, scales, rotations, vs.scale_modifier, cov3Ds_precomp, vs.nVoxel_x, vs.nVoxel_y,
                vs.nVoxel_z, vs.sVoxel_x, vs.sVoxel_y, vs.sVoxel_z, vs.center_x, vs.center_y, vs.center_z,
                vs.prefiltered, vs.debug)
        slab = _slab_of(vs)
        (num_rendered, fields, radii_x, radii_y, radii_z, geomBuffer, binningBuffer, imgBuffer) = _guarded(
            _C.voxelize_gaussians if slab is None else _C.voxelize_gaussians_slab, args if slab is None else args + slab,
            vs.debug, "snapshot_fw.dump", "forward")
        ctx.voxel_settings = vs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(means3D, scales, rotations, cov3Ds_precomp, radii_x, radii_y, radii_z, geomBuffer,
                              binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii_x, radii_y, radii_z)
        ctx.set_materialize_grads(False)   # no zero tensors for the unused radii gradient slots
        return fields, radii_x, radii_y, radii_z

    @staticmethod
    def backward(ctx, grad_out_color, _gx, _gy, _gz):
        vs = ctx.voxel_settings
        (means3D, scales, rotations, cov3Ds_precomp, radii_x, radii_y, radii_z, geomBuffer, binningBuffer,
         imgBuffer) = ctx.saved_tensors
        if grad_out_color is None:   # the volume did not take part in the loss
            return None, None, None, None, None, None
        args = (means3D, radii_x, radii_y, radii_z, scales, rotations, vs.scale_modifier, cov3Ds_precomp,
                grad_out_color, geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, vs.nVoxel_x, vs.nVoxel_y,
                vs.nVoxel_z, vs.sVoxel_x, vs.sVoxel_y, vs.sVoxel_z, vs.center_x, vs.center_y, vs.center_z, vs.debug)
        slab = _slab_of(vs)
        grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_scales, grad_rotations = _guarded(
            _C.voxelize_gaussians_backward if slab is None else _C.voxelize_gaussians_backward_slab,
            args if slab is None else args + slab, vs.debug, "snapshot_bw.dump", "backward")
        if scales.numel() == 0:
            grad_scales = None
        if rotations.numel() == 0:
            grad_rotations = None
        if cov3Ds_precomp.numel() == 0:
            grad_cov3Ds_precomp = None
        return grad_means3D, grad_opacities, grad_scales, grad_rotations, grad_cov3Ds_precomp, None


def voxelize_gaussians(means3D, opacities, scales, rotations, cov3Ds_precomp, voxel_settings):
    fields, rx, ry, rz = _VoxelizeGaussians.apply(means3D, opacities, scales, rotations, cov3Ds_precomp,
                                                  voxel_settings)
    return fields, (rx, ry, rz)


class GaussianVoxelizer(nn.Module):
    def __init__(self, voxel_settings):
        super().__init__()
        self.voxel_settings = voxel_settings

    def forward(self, means3D, opacities, scales=None, rotations=None, cov3D_precomp=None):
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception(
                "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if scales is None:
            scales = torch.Tensor([])
        if rotations is None:
            rotations = torch.Tensor([])
        if cov3D_precomp is None:
            cov3D_precomp = torch.Tensor([])
        return voxelize_gaussians(means3D, opacities, scales, rotations, cov3D_precomp, self.voxel_settings)
