"""distCUDA2 on the GPU vs the exhaustive CPU oracle: bit-exact (same float op order, order-independent 3-best)."""
import numpy as np
import pytest
import torch

from r2_gaussian_amd import scene as S

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P", [4, 7, 1000, 1025, 20000])
def test_knn_bit_exact(P, oracle, gpu):
    from r2_gaussian_amd import distCUDA2
    pts = S.make_cloud(P, seed=P).xyz
    ref = oracle.knn_dist2(pts.numpy())
    out = distCUDA2(pts.to(gpu)).cpu().numpy()
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    assert (out > 0).all()


def test_knn_duplicates_and_tiny(oracle, gpu):
    from r2_gaussian_amd import distCUDA2
    pts = S.make_cloud(64, seed=1).xyz
    pts[5] = pts[4]          # a duplicate is a neighbour at distance 0 (only the query itself is excluded)
    ref = oracle.knn_dist2(pts.numpy())
    out = distCUDA2(pts.to(gpu)).cpu().numpy()
    assert np.array_equal(out, ref)
    assert distCUDA2(torch.zeros((0, 3), device=gpu)).numel() == 0
    # fewer than 4 points: missing neighbours stay at FLT_MAX, as in the upstream kernel (sum overflows to inf)
    two = distCUDA2(pts[:2].to(gpu)).cpu().numpy()
    assert np.array_equal(two, oracle.knn_dist2(pts[:2].numpy()))


def test_create_from_pcd_usage(oracle, gpu):
    """The one call site: dist = sqrt(clamp_min(distCUDA2(xyz), 1e-6)) (gaussian_model.py:145-150)."""
    from simple_knn._C import distCUDA2
    xyz = S.make_cloud(5000, seed=9).xyz.to(gpu)
    dist = torch.sqrt(torch.clamp_min(distCUDA2(xyz), 0.001 ** 2))
    assert dist.shape == (5000,) and torch.isfinite(dist).all() and (dist > 0).all()
