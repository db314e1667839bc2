"""The fused loss kernels (csrc/loss_ops.hip) against a plain PyTorch fp32 evaluation of the reference's loss functions
(r2_gaussian/utils/loss_utils.py:19-104, restated in tests/mini_trainer.py) on the CPU: values and gradients."""
import numpy as np
import pytest
import torch

from tests import mini_trainer as T

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hw", [(64, 64), (50, 70), (128, 96), (512, 512)], ids=["64", "ragged_50x70", "128x96", "512"])
def test_l1_ssim_value_and_gradient(hw, gpu):
    from r2_gaussian_amd.losses import image_loss
    g = torch.Generator().manual_seed(hw[0])
    gt = torch.rand(1, *hw, generator=g)
    img = (gt + 0.1 * torch.randn(1, *hw, generator=g)).clamp_min(0.0)
    img[0, :4, :4] = gt[0, :4, :4]          # exact zeros of x - y: sign(0) = 0 like torch
    a = img.clone().requires_grad_(True)
    ref = (a - gt).abs().mean() + 0.25 * (1.0 - T.ssim(a, gt))
    ref.backward()
    b = img.clone().to(gpu).requires_grad_(True)
    loss, parts = image_loss(b, gt.to(gpu), 0.25)
    (loss * 2.0).backward()                  # upstream gradient 2
    torch.cuda.synchronize()
    assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref)) + 1e-7
    assert abs(float(parts[0]) - float((img - gt).abs().mean())) <= 1e-6
    assert abs(float(parts[1]) - float(T.ssim(img, gt))) <= 2e-6
    got, want = b.grad.cpu().numpy() / 2.0, a.grad.numpy()
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 2e-5 * scale, np.abs(got - want).max() / scale


@pytest.mark.parametrize("n", [(32, 32, 32), (8, 20, 13)], ids=["tv_32cube", "ragged"])
def test_tv3d_value_and_gradient(n, gpu):
    from r2_gaussian_amd.losses import tv_3d_loss
    g = torch.Generator().manual_seed(5)
    vol = torch.rand(*n, generator=g)
    vol[1, 2, 3] = vol[1, 2, 4]              # a zero difference
    a = vol.clone().requires_grad_(True)
    ref = T.tv3d_mean(a)
    ref.backward()
    b = vol.clone().to(gpu).requires_grad_(True)
    tv = tv_3d_loss(b)
    (0.05 * tv).backward()
    torch.cuda.synchronize()
    assert abs(float(tv) - float(ref)) <= 2e-6 * float(ref)
    np.testing.assert_allclose(b.grad.cpu().numpy() / 0.05, a.grad.numpy(), rtol=1e-5, atol=1e-9)


def test_trainer_with_fused_losses_follows_the_torch_losses(gpu):
    """The miniature trainer with the fused loss nodes in place of the torch loss stack: same trajectory (the losses differ
    in the last bits only)."""
    case = T.Case(detector=64, n_vol=32, n_views=10, p_gt=2000, n_init=1500, seed=2)
    opt = dict(iterations=120, densify_from_iter=40, densify_until_iter=100, densification_interval=20)
    a = T.train(case, T.Opt(**opt), "hip", eval_every=40, seed=0)
    b = T.train(case, T.Opt(**opt), "hip", eval_every=40, seed=0, fused_losses=True)
    assert a["iters"] == b["iters"]
    assert max(abs(x - y) for x, y in zip(a["psnr"], b["psnr"])) < 0.05
    assert b["psnr"][-1] > b["psnr"][0] + 1.0


# ---- against the reference's own Python (tests/golden/train/losses.npz, written by loss_utils.py itself: make_golden_train.py)
def test_fused_losses_match_the_reference_fixtures(gpu):
    import os
    from r2_gaussian_amd.losses import image_loss, tv_3d_loss
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "train", "losses.npz"))
    lam = 0.25
    for i in range(3):
        img = torch.from_numpy(g["img%d" % i]).to(gpu).requires_grad_(True)
        gt = torch.from_numpy(g["gt%d" % i]).to(gpu)
        loss, parts = image_loss(img, gt, lam)
        loss.backward()
        torch.cuda.synchronize()
        l1, ssim = float(g["l1_%d" % i]), float(g["ssim_%d" % i])
        assert abs(float(parts[0]) - l1) <= 1e-6 * abs(l1) + 1e-8
        assert abs(float(parts[1]) - ssim) <= 2e-6
        assert abs(float(loss) - (l1 + lam * (1.0 - ssim))) <= 2e-6
        want = g["l1_grad%d" % i] - lam * g["ssim_grad%d" % i]       # d/dimg of L1 + lam (1 - SSIM)
        got = img.grad.cpu().numpy()
        assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max()
    for i in range(2):
        vol = torch.from_numpy(g["vol%d" % i]).to(gpu).requires_grad_(True)
        tv = tv_3d_loss(vol)
        tv.backward()
        torch.cuda.synchronize()
        assert abs(float(tv) - float(g["tv_%d" % i])) <= 2e-6 * float(g["tv_%d" % i])
        want = g["tv_grad%d" % i]
        assert np.abs(vol.grad.cpu().numpy() - want).max() <= 1e-6 * np.abs(want).max() + 1e-12
