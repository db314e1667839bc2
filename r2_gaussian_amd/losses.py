"""The training loop's loss stack on the MI355X kernels (csrc/loss_ops.hip): drop-in replacements for
``l1_loss + lambda_dssim * (1 - ssim)`` and ``tv_3d_loss(vol, "mean")`` of r2_gaussian/utils/loss_utils.py, each as ONE
autograd node whose forward already computes the gradient (two / one kernel launches instead of ~60 torch kernels, and no
vendor convolution).  The results are device scalars: nothing here synchronises with the host (train.py:204-209 calls
``.item()`` on every loss every iteration; read the tensors only when something is logged).
"""
import torch

from . import _lib
from ._C import _on_device, _require_gpu, _stream

_F32 = torch.float32


class _ImageLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, w_l1, w_ssim):
        _require_gpu(image, "image")
        img = image.reshape(image.shape[-2], image.shape[-1]).to(_F32).contiguous()
        ref = gt.reshape(gt.shape[-2], gt.shape[-1]).to(device=img.device, dtype=_F32).contiguous()
        H, W = img.shape
        L = _lib.lib()
        grad = torch.empty_like(img)
        scratch = torch.empty(L.r2_loss_l1_ssim_scratch_floats(W, H), dtype=_F32, device=img.device)
        scalars = torch.empty(3, dtype=_F32, device=img.device)
        with _on_device(img.device):
            rc = L.r2_loss_l1_ssim(W, H, img.data_ptr(), ref.data_ptr(), float(w_l1), float(w_ssim), grad.data_ptr(),
                                   scratch.data_ptr(), scalars.data_ptr(), _stream(img.device))
        _lib.check(rc, "r2_loss_l1_ssim")
        ctx.save_for_backward(grad)
        ctx.shape = image.shape
        ctx.mark_non_differentiable(scalars)
        return scalars[2], scalars

    @staticmethod
    def backward(ctx, g, _):
        (grad,) = ctx.saved_tensors
        return (grad * g).reshape(ctx.shape), None, None, None


def image_loss(image, gt, lambda_dssim=0.25):
    """-> (loss, parts): loss = L1 + lambda_dssim * (1 - SSIM) as a device scalar with a gradient; parts = tensor
    {l1, ssim, loss} for logging (train.py:118-126)."""
    return _ImageLoss.apply(image, gt, 1.0, float(lambda_dssim))


class _TV3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vol):
        _require_gpu(vol, "vol")
        v = vol.to(_F32).contiguous()
        nx, ny, nz = v.shape
        L = _lib.lib()
        grad = torch.empty_like(v)
        scratch = torch.empty(L.r2_loss_tv3d_scratch_floats(nx, ny, nz), dtype=_F32, device=v.device)
        scalars = torch.empty(2, dtype=_F32, device=v.device)
        with _on_device(v.device):
            rc = L.r2_loss_tv3d(nx, ny, nz, v.data_ptr(), 1.0, grad.data_ptr(), scratch.data_ptr(), scalars.data_ptr(),
                                _stream(v.device))
        _lib.check(rc, "r2_loss_tv3d")
        ctx.save_for_backward(grad)
        return scalars[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g


def tv_3d_loss(vol):
    """tv_3d_loss(vol, reduction="mean") of loss_utils.py:19-34 as one autograd node."""
    return _TV3D.apply(vol)
