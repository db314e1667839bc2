"""Row f4 on the MI355X: the FDK kernels (csrc/fdk.hip) against the float64 restatement (oracle/fdk_oracle.py), and against
physics -- projections of a known density rendered by the X-ray rasterizer must reconstruct to the voxelizer's volume of that
density, in the voxelizer's orientation.  Tolerance: 1e-4 of the sum of the magnitudes of the terms of each output (filter: a
row's |x| * |taps|; back-projection: the per-voxel sum over the views of |weighted sample|)."""
import json
import os
import time

import numpy as np
import pytest
import torch

from oracle import fdk_oracle as F
from r2_gaussian_amd import fdk as K
from r2_gaussian_amd import scene as S
from tests import helpers as Hh
from tests.test_fdk_cpu import ball_projection

pytestmark = pytest.mark.gpu
RTOL = 1e-4


@pytest.mark.parametrize("name", F.FILTERS)
@pytest.mark.parametrize("W,H,cone", [(64, 64, True), (100, 37, True), (130, 50, False), (1030, 9, True), (2100, 5, True)])
def test_filter_matches_the_fft_formulation(gpu, name, W, H, cone):
    if W > 1024 and name != "hann":
        pytest.skip("long rows: one window is enough")
    rng = np.random.RandomState(W)
    p = rng.rand(3, H, W).astype(np.float32)
    du, dv = 0.013, 0.017
    ref = F.fdk_filter(p, du, dv, 7.0, 5.0, name, cone)
    got = K.fdk_filter(torch.from_numpy(p).to(gpu), du, dv, 7.0, 5.0, name, cone).cpu().numpy()
    assert got.shape == (3, W, H)
    got = got.transpose(0, 2, 1)
    t = np.abs(F.spatial_taps(W, H, name))
    bound = np.stack([[np.convolve(np.abs(p[v, r]), t)[W - 1:2 * W - 1] for r in range(H)] for v in range(3)])
    bound *= F.filter_scale(3, du, 7.0, 5.0, cone)
    err = np.abs(got - ref)
    assert (err <= RTOL * bound + 1e-30).all(), float((err / (bound + 1e-30)).max())
    Hh._log("fdk", "filter %s W=%d H=%d cone=%d" % (name, W, H, cone), lambda: {"max_err_over_tol": float((err / (RTOL * bound + 1e-30)).max())})


@pytest.mark.parametrize("cone", [True, False])
def test_backprojection_matches_the_restatement(gpu, cone):
    scanner = S.CONE_BEAM if cone else S.PARALLEL_BEAM
    H, W, V = 80, 96, 40
    views = S.make_views(V, (H, W), scanner)
    rng = np.random.RandomState(3)
    q = (rng.rand(V, H, W) - 0.5).astype(np.float32)
    fp = torch.stack([v.full_proj_transform for v in views])
    n, s, ctr = (24, 20, 28), (2.0, 1.6, 2.2), (0.05, -0.1, 0.02)
    ref, absum = F.fdk_backproject(q, fp.numpy(), 5.0, n, s, ctr, cone, return_abs=True)
    q_t = torch.from_numpy(np.ascontiguousarray(q.transpose(0, 2, 1))).to(gpu)
    got = K.fdk_backproject(q_t, fp, 5.0, n, s, ctr, cone).cpu().numpy()
    err = np.abs(got - ref)
    # the detector coordinate is a float32 of magnitude W: 3e-5 pixel of quantisation under a slope of up to |q| per pixel
    assert (err <= RTOL * absum + 1e-30).all(), float((err / (absum + 1e-30)).max())
    Hh._log("fdk", "backproject cone=%d" % cone, lambda: {"max_err_over_tol": float((err / (RTOL * absum + 1e-30)).max())})
    # twice the same bits
    again = K.fdk_backproject(q_t, fp, 5.0, n, s, ctr, cone).cpu().numpy()
    assert np.array_equal(got, again)


def test_ball_reconstructs_to_its_density(gpu):
    n, V = 128, 180
    cfg = dict(S.CONE_BEAM, nVoxel=[64, 64, 64], filter=None)
    views = S.make_views(V, (n, n))
    projs = np.repeat(ball_projection(n, 0.5, views[0])[None], V, 0).astype(np.float32)
    vol = K.fdk(projs, [v.angle for v in views], cfg, device=gpu).cpu().numpy()
    assert abs(vol[28:36, 28:36, 28:36].mean() - 1.0) < 0.01
    assert np.abs(vol[32, :8, 32]).max() < 0.03
    # and it is the restatement's volume
    fp = np.stack([v.full_proj_transform.numpy() for v in views])
    ref = F.fdk(projs, fp, 4.0 / n, 4.0 / n, 7.0, 5.0, (64, 64, 64), (2, 2, 2), (0, 0, 0))
    assert np.abs(vol - ref).max() < 2e-3   # sum of 180 terms of magnitude ~1 each (|filtered| ~ 30 x (pi/180) weights)


def _corr(a, b):
    a, b = a - a.mean(), b - b.mean()
    return float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum()))


@pytest.mark.parametrize("scanner", [S.CONE_BEAM, S.PARALLEL_BEAM], ids=["cone", "parallel"])
def test_rasterizer_projections_reconstruct_to_the_voxelizer_volume(gpu, oracle, scanner):
    """The physical anchor that stands in for TIGRE's output: FDK(X-ray projections of a cloud) ~ voxelisation of the cloud,
    same orientation, same amplitude -- for both beam geometries of the reference."""
    c = S.make_cloud(4000, seed=7, scale_mult=2.5)
    n_det, V = 128, 180
    views = S.make_views(V, (n_det, n_det), scanner)
    projs = torch.stack([torch.as_tensor(Hh.hip_raster(c, v, gpu)["color"]).reshape(n_det, n_det) for v in views])
    cfg = dict(scanner, nVoxel=[64, 64, 64], filter=None)
    vol = K.fdk(projs.to(gpu), [v.angle for v in views], cfg).cpu().numpy()
    truth = Hh.hip_voxel(c, (64, 64, 64), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0), gpu)["vol"]
    r = _corr(vol, truth)
    gain = float((vol * truth).sum() / (truth * truth).sum())
    psnr = S.psnr3d(torch.as_tensor(truth), torch.as_tensor(vol), pixel_max=float(truth.max()))
    Hh._log("fdk", "raster->fdk vs voxelizer (%s)" % scanner["mode"], lambda: {"corr": r, "gain": gain, "psnr_db": psnr})
    if scanner["mode"] == "cone":
        assert r > 0.999 and abs(gain - 1.0) < 0.01 and psnr > 40.0, (r, gain, psnr)   # CPU restatement: 0.99977, 0.9955, 45.3 dB
    else:
        # the orthographic detector spans exactly the volume's [-1, 1]: the cloud's 5 % of points in the cube's corners leave
        # it at oblique angles (truncated projections), which costs the border -- CPU restatement: 0.99539, 1.00256, 31.4 dB
        # overall, and 0.999997 / 60.5 dB inside [12:52]^3
        inner = slice(12, 52)
        ri = _corr(vol[inner, inner, inner], truth[inner, inner, inner])
        pi_ = S.psnr3d(torch.as_tensor(truth[inner, inner, inner]), torch.as_tensor(vol[inner, inner, inner]), pixel_max=float(truth.max()))
        assert r > 0.99 and abs(gain - 1.0) < 0.01 and ri > 0.9999 and pi_ > 50.0, (r, gain, psnr, ri, pi_)
    for flipped in (vol[::-1], vol[:, ::-1], vol[:, :, ::-1], vol.transpose(1, 0, 2), vol.transpose(2, 1, 0)):
        assert _corr(np.ascontiguousarray(flipped), truth) < r - 0.005   # a dense blob: mirrored copies still reach 0.93-0.99
    # init_pcd on top: the sampled centres lie where the density is, their densities are the rescaled FDK values
    pts = K.init_pcd(projs.cpu().numpy(), [v.angle for v in views], cfg, n_points=2000, density_thresh=0.02,
                     rng=np.random.RandomState(0))
    assert pts.shape == (2000, 4)
    idx = np.rint((pts[:, :3] + 1.0) / (2.0 / 64)).astype(int)
    assert np.allclose(pts[:, 3], vol[idx[:, 0], idx[:, 1], idx[:, 2]] * 0.15, rtol=1e-6)
    assert (truth[idx[:, 0], idx[:, 1], idx[:, 2]] > 0.005).mean() > (0.95 if scanner["mode"] == "cone" else 0.9)


def test_headline_size_timing(gpu):
    """512^2 x 50 views -> 256^3 (the benchmark scene's initialisation) and 1024^2 x 360 -> 256^3: time both kernels."""
    out = {}
    for (n_det, V) in ((512, 50), (1024, 360)):
        views = S.make_views(V, (n_det, n_det))
        p = torch.rand(V, n_det, n_det, device=gpu)
        fp = torch.stack([v.full_proj_transform for v in views])
        du = 4.0 / n_det
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            q = K.fdk_filter(p, du, du, 7.0, 5.0, "ram_lak")
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            vol = K.fdk_backproject(q, fp, 5.0, (256, 256, 256), (2, 2, 2), (0, 0, 0))
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        assert torch.isfinite(vol).all()
        out["%dx%d_%dviews" % (n_det, n_det, V)] = {
            "filter_ms": (t1 - t0) * 1e3, "backproject_ms": (t2 - t1) * 1e3,
            "filter_tflops": 2.0 * V * n_det * n_det * n_det / (t1 - t0) / 1e12,
            "backproject_gupdates_per_s": V * 256.0 ** 3 / (t2 - t1) / 1e9}
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/fdk_timing.json", "w"), indent=1)
    print(out)


def test_argument_errors(gpu):
    from r2_gaussian_amd._lib import R2HipError
    with pytest.raises(R2HipError, match="4096"):
        K.fdk_filter(torch.zeros(1, 2, 4100, device=gpu), 0.01, 0.01, 7.0, 5.0)
    with pytest.raises(ValueError):
        K.fdk_filter(torch.zeros(1, 8, 8, device=gpu), 0.01, 0.01, 7.0, 5.0, "butterworth")
    with pytest.raises(R2HipError):
        K.fdk_filter(torch.zeros(1, 8, 8), 0.01, 0.01, 7.0, 5.0)   # CPU tensor: no fallback
