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


def _grid_vs_exhaustive(pts, gpu, tmp_env=None):
    """distCUDA2 through the grid search (P >= 4096) vs the exhaustive kernel of the same library (a sub-4096 call cannot be
    forced, so the exhaustive result comes from a subprocess with R2_KNN_GRID=0)."""
    import os
    import subprocess
    import sys
    import tempfile
    from r2_gaussian_amd import distCUDA2
    got = distCUDA2(pts.to(gpu)).cpu().numpy()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "p.npy"), pts.numpy())
        code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from r2_gaussian_amd import distCUDA2; "
                "p = torch.from_numpy(np.load(%r)).cuda(); np.save(%r, distCUDA2(p).cpu().numpy())"
                % (root, os.path.join(d, "p.npy"), os.path.join(d, "o.npy")))
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, R2_KNN_GRID="0"), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        ref = np.load(os.path.join(d, "o.npy"))
    return got, ref


@pytest.mark.parametrize("kind", ["cloud-50k", "cloud-300k", "lattice", "clustered", "flat", "collinear", "duplicates", "identical"])
def test_knn_grid_search_is_bit_identical_to_the_exhaustive_one(kind, oracle, gpu):
    """Round 4: P >= 4096 points take a uniform-grid search (csrc/knn.hip) -- the same 3-best multiset, hence the same bits, as the
    exhaustive kernel (and as the oracle's brute force where that is affordable): a cloud, the voxel lattice initialize_pcd.py
    samples from (many exactly equal distances), heavy clustering (long cells, many rings for the outliers), a plane, a line,
    exact duplicates, all points identical (no grid possible: the exhaustive kernel serves it)."""
    g = torch.Generator().manual_seed(3)
    if kind == "cloud-50k":
        pts = S.make_cloud(50000, seed=4).xyz
    elif kind == "cloud-300k":
        pts = S.make_cloud(300000, seed=5).xyz
    elif kind == "lattice":
        idx = torch.randperm(64 ** 3, generator=g)[:40000]
        pts = torch.stack([idx // 4096, (idx // 64) % 64, idx % 64], 1).float() * (2.0 / 64) - 1.0
    elif kind == "clustered":
        pts = torch.cat([torch.randn(30000, 3, generator=g) * 0.003 + 0.4, torch.rand(2000, 3, generator=g) * 2 - 1,
                         torch.randn(8000, 3, generator=g) * 0.01 - 0.5])
    elif kind == "flat":
        pts = torch.rand(20000, 3, generator=g) * 2 - 1
        pts[:, 2] = 0.25
    elif kind == "collinear":
        pts = torch.zeros(10000, 3)
        pts[:, 0] = torch.rand(10000, generator=g)
    elif kind == "duplicates":
        base = S.make_cloud(3000, seed=6).xyz
        pts = base.repeat(3, 1)
    else:
        pts = torch.full((5000, 3), 0.125)
    pts = pts.contiguous()
    got, ref = _grid_vs_exhaustive(pts, gpu)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), kind
    if pts.shape[0] <= 50000:
        assert np.array_equal(got.view(np.uint32), oracle.knn_dist2(pts.numpy()).view(np.uint32)), kind


@pytest.mark.parametrize("kind", ["far-from-origin", "elongated-512x8x8", "one-dense-blob", "workspace-too-small"])
def test_knn_grid_stop_test_and_fallbacks(kind, oracle, gpu):
    """Round 5.  (1) The ring search stops from a bound derived from the cell FUNCTION (which cell can a point at this
    distance still fall into, with its two float roundings), not from the nominal cell walls shrunk by an ad hoc margin: a cloud
    translated by 100x its extent, and one 512 cells long and 8 across, are where the old margin was of the size of the error.
    (2) A cloud whose points crowd into one cell (a query there would scan O(P) points per lane) is handed to the exhaustive
    kernel.  (3) The search works in a caller-provided workspace (r2_knn_workspace_bytes); without one the exhaustive kernel
    runs.  All of them: the exhaustive kernel's bits."""
    g = torch.Generator().manual_seed(11)
    if kind == "far-from-origin":
        pts = S.make_cloud(30000, seed=8).xyz + torch.tensor([200.0, -150.0, 120.0])
    elif kind == "elongated-512x8x8":
        pts = torch.rand(60000, 3, generator=g) * torch.tensor([64.0, 1.0, 1.0]) + torch.tensor([30.0, 0.0, -3.0])
    elif kind == "one-dense-blob":
        pts = torch.cat([torch.randn(20000, 3, generator=g) * 1e-5 + 0.3, torch.rand(4000, 3, generator=g) * 2 - 1])
    else:
        pts = S.make_cloud(9000, seed=3).xyz
    pts = pts.contiguous()
    if kind == "workspace-too-small":
        from r2_gaussian_amd import _lib
        from r2_gaussian_amd._C import _stream
        L = _lib.lib()
        need = int(L.r2_knn_workspace_bytes(pts.shape[0]))
        assert need > 16 * pts.shape[0]
        dp = pts.to(gpu)
        outs = []
        for nbytes in (need, need // 2, 0):     # full workspace -> grid search; too small / none -> exhaustive kernel
            ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=gpu)
            o = torch.zeros(pts.shape[0], device=gpu)
            rc = L.r2_knn_dist2_ws(pts.shape[0], dp.data_ptr(), o.data_ptr(), ws.data_ptr() if nbytes else None, nbytes, _stream(gpu))
            assert rc == 0, L.r2_last_error()
            torch.cuda.synchronize()
            outs.append(o.cpu().numpy())
        ref = oracle.knn_dist2(pts.numpy())
        for o in outs:
            assert np.array_equal(o.view(np.uint32), ref.view(np.uint32))
        return
    got, ref = _grid_vs_exhaustive(pts, gpu)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), kind
    if pts.shape[0] <= 30000:
        assert np.array_equal(got.view(np.uint32), oracle.knn_dist2(pts.numpy()).view(np.uint32)), kind

