"""Drop-in for ``xray_gaussian_rasterization_voxelization.rasterization``
(SUB/xray_gaussian_rasterization_voxelization/rasterization.py): same public names, argument order,
return values, autograd contract and error behaviour; the compute goes to the MI355X kernels via ``_C``.

Autograd contract (PY/rasterization.py:186-196): ``backward(grad_color, _)`` returns gradients for
``(means3D, means2D, opacities, scales, rotations, cov3Ds_precomp, None)``; the ``means2D`` gradient is in
NDC units and feeds the densification statistics (r2_gaussian/gaussian/gaussian_model.py:552-556).
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


def cpu_deep_copy_tuple(input_tuple):
    return tuple(x.cpu().clone() if isinstance(x, torch.Tensor) else x for x in input_tuple)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    campos: torch.Tensor
    prefiltered: bool
    mode: int
    debug: bool


def _guarded(fn, args, debug, dump_name, where):
    """debug mode: keep a CPU copy of the arguments and dump it if the native call throws
    (PY/rasterization.py:80-93,156-175)."""
    if not debug:
        return fn(*args)
    cpu_args = cpu_deep_copy_tuple(args)
    try:
        return fn(*args)
    except Exception:
        torch.save(cpu_args, dump_name)
        print("\nAn error occured in %s. Writing %s for debugging.\n" % (where, dump_name))
        raise


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        rs = raster_settings
        args = (means3D, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
                rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, rs.campos, rs.prefiltered,
                rs.mode, rs.debug)
        num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = _guarded(
            _C.rasterize_gaussians, args, rs.debug, "snapshot_fw.dump", "forward")
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.mode = rs.mode
        ctx.save_for_backward(means3D, scales, rotations, cov3Ds_precomp, radii, geomBuffer, binningBuffer,
                              imgBuffer)
        ctx.mark_non_differentiable(radii)
        # the gradient slot of `radii` is never used: do not let autograd materialise a zero int tensor (one fill kernel
        # per backward) for it
        ctx.set_materialize_grads(False)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _):
        rs = ctx.raster_settings
        means3D, scales, rotations, cov3Ds_precomp, radii, geomBuffer, binningBuffer, imgBuffer = ctx.saved_tensors
        if grad_out_color is None:   # the image did not take part in the loss
            return None, None, None, None, None, None, None
        args = (means3D, radii, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix, rs.projmatrix,
                rs.tanfovx, rs.tanfovy, grad_out_color, rs.campos, geomBuffer, ctx.num_rendered, binningBuffer,
                imgBuffer, ctx.mode, rs.debug)
        (grad_means2D, grad_opacities, _grad_mu, grad_means3D, grad_cov3Ds_precomp, grad_scales,
         grad_rotations) = _guarded(_C.rasterize_gaussians_backward, args, rs.debug, "snapshot_bw.dump", "backward")
        # inputs that were passed as empty placeholders get no gradient
        if scales.numel() == 0:
            grad_scales = None
        if rotations.numel() == 0:
            grad_rotations = None
        if cov3Ds_precomp.numel() == 0:
            grad_cov3Ds_precomp = None
        return grad_means3D, grad_means2D, grad_opacities, grad_scales, grad_rotations, grad_cov3Ds_precomp, None


def rasterize_gaussians(means3D, means2D, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """bool[P]: in front of the near plane of this view (PY/rasterization.py:219-227)."""
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, scales=None, rotations=None, cov3D_precomp=None):
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
        return rasterize_gaussians(means3D, means2D, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings)


# ---------------------------------------------------------------------------------------------- batched views (new)
class _RasterizeGaussiansBatch(torch.autograd.Function):
    """V views of the same Gaussians in one pass (``_C.rasterize_gaussians_batch``): ``raster_settings.viewmatrix`` /
    ``.projmatrix`` are [V,4,4]; returns color [V,H,W], radii [V,P].  ``means2D`` is [V,P,3] (one screen-space gradient
    holder per view: the densification statistics are per view); the parameter gradients are the sum over the views."""

    @staticmethod
    def forward(ctx, means3D, means2D, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        rs = raster_settings
        num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = _C.rasterize_gaussians_batch(
            means3D, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix, rs.projmatrix,
            rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, rs.mode, rs.debug)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(means3D, scales, rotations, cov3Ds_precomp, radii, geomBuffer, binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _):
        rs = ctx.raster_settings
        means3D, scales, rotations, cov3Ds_precomp, radii, geomBuffer, binningBuffer, imgBuffer = ctx.saved_tensors
        if grad_out_color is None:
            return None, None, None, None, None, None, None
        (grad_means2D, grad_opacities, _grad_mu, grad_means3D, grad_cov3Ds_precomp, grad_scales,
         grad_rotations) = _C.rasterize_gaussians_backward_batch(
            means3D, radii, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
            rs.tanfovy, grad_out_color, geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, rs.mode, rs.debug)
        if scales.numel() == 0:
            grad_scales = None
        if rotations.numel() == 0:
            grad_rotations = None
        if cov3Ds_precomp.numel() == 0:
            grad_cov3Ds_precomp = None
        return grad_means3D, grad_means2D, grad_opacities, grad_scales, grad_rotations, grad_cov3Ds_precomp, None


class GaussianRasterizerBatch(nn.Module):
    """``GaussianRasterizer`` for V views at once: settings with viewmatrix / projmatrix of shape [V,4,4]."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, scales=None, rotations=None, cov3D_precomp=None):
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception(
                "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        e = torch.Tensor([])
        return _RasterizeGaussiansBatch.apply(means3D, means2D, opacities, e if scales is None else scales,
                                              e if rotations is None else rotations,
                                              e if cov3D_precomp is None else cov3D_precomp, self.raster_settings)
