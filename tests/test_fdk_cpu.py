"""Row f4 on the CPU: the FDK restatement (oracle/fdk_oracle.py) reconstructs an analytic phantom to its density, its
spatial-tap form of TIGRE's FFT filter is the same operator, the product's host-side filter design
(r2_gaussian_amd/fdk.py:ramp_taps) agrees with it, and the point sampling of init_pcd equals the reference's statements
(initialize_pcd.py:63-86) draw for draw."""
import numpy as np
import pytest

from oracle import fdk_oracle as F
from r2_gaussian_amd import fdk as K
from r2_gaussian_amd import scene as S


def ball_projection(n, radius, view):
    """Line integrals of a unit-density ball at the origin on an n x n detector (the same for every angle)."""
    c = (np.arange(n) + 0.5) * 2.0 / n - 1.0
    DY, DX = np.meshgrid(c * view.tanfovy, c * view.tanfovx, indexing="ij")
    d = np.stack([DX, DY, np.ones_like(DX)], -1)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    ctr = np.array([0.0, 0.0, 5.0])
    b = d @ ctr
    return 2.0 * np.sqrt(np.maximum(radius * radius - (ctr @ ctr - b * b), 0.0))


def test_fdk_oracle_reconstructs_a_ball_to_its_density():
    n, V = 96, 120
    views = S.make_views(V, (n, n))
    projs = np.repeat(ball_projection(n, 0.5, views[0])[None], V, 0)
    fp = np.stack([v.full_proj_transform.numpy() for v in views])
    vol = F.fdk(projs, fp, 4.0 / n, 4.0 / n, 7.0, 5.0, (48, 48, 48), (2, 2, 2), (0, 0, 0))
    assert abs(vol[20:28, 20:28, 20:28].mean() - 1.0) < 0.01          # inside: the density
    assert np.abs(vol[24, :6, 24]).max() < 0.03 and np.abs(vol[:6, 24, 24]).max() < 0.03   # outside: ~0
    # the ball's edge sits at voxel 12 / 36 along every axis through the centre (r = 0.5 = 12 voxels)
    for line in (vol[:, 24, 24], vol[24, :, 24], vol[24, 24, :]):
        assert line[10] < 0.1 and line[13] > 0.9 and line[34] > 0.9 and line[37] < 0.1


@pytest.mark.parametrize("name", F.FILTERS)
@pytest.mark.parametrize("W,H", [(64, 64), (100, 37), (130, 200)])
def test_spatial_taps_are_tigres_fft_filter(name, W, H):
    rng = np.random.RandomState(W + H)
    p = rng.rand(3, H, W)
    du, dv = 0.013, 0.017
    ref = F.fdk_filter(p, du, dv, 7.0, 5.0, name)
    t = F.spatial_taps(W, H, name)
    x = p * F.preweight(H, W, dv, du, 7.0)[None]
    got = np.stack([[np.convolve(x[v, r], t)[W - 1:2 * W - 1] for r in range(H)] for v in range(3)])
    got *= F.filter_scale(3, du, 7.0, 5.0)
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    # the product's design of the same taps (float32)
    tk = K.ramp_taps(W, H, name)
    assert tk.dtype == np.float32 and tk.shape == (2 * W - 1,)
    assert np.abs(tk - t).max() <= 1e-7 * np.abs(t).max()


def test_ram_lak_taps_closed_form():
    W = 50
    t = F.spatial_taps(W, W, "ram_lak")
    n = np.arange(-(W - 1), W)
    want = np.where(n == 0, 0.5, np.where(n % 2 != 0, -2.0 / (np.pi * np.where(n == 0, 1, n)) ** 2, 0.0))
    assert np.abs(t - want).max() < 1e-12
    with pytest.raises(ValueError):
        K.ramp_taps(32, 32, "butterworth")


def test_init_pcd_sampling_follows_the_reference(monkeypatch):
    cfg = dict(S.CONE_BEAM, nVoxel=[20, 24, 28], filter=None)
    rs = np.random.RandomState(5)
    vol = rs.rand(20, 24, 28).astype(np.float32) * 0.2
    monkeypatch.setattr(K, "recon_volume", lambda projs, angles, c, m: vol)
    out = K.init_pcd(None, None, cfg, n_points=500, density_thresh=0.05, density_rescale=0.15,
                     rng=np.random.RandomState(0))
    # initialize_pcd.py:66-86 with np.random.seed(0)
    np.random.seed(0)
    valid = np.argwhere(vol > 0.05)
    idx = valid[np.random.choice(len(valid), 500, replace=False)]
    dV, sV = np.array(cfg["sVoxel"]) / np.array(cfg["nVoxel"]), np.array(cfg["sVoxel"])
    pos = idx * dV - sV / 2 + np.array(cfg["offOrigin"])
    dens = vol[idx[:, 0], idx[:, 1], idx[:, 2]] * 0.15
    assert np.array_equal(out, np.concatenate([pos, dens[:, None]], -1))
    # random mode (initialize_pcd.py:49-58)
    out = K.init_pcd(None, None, cfg, n_points=100, recon_method="random", rng=np.random.RandomState(0))
    np.random.seed(0)
    pos = np.array(cfg["offOrigin"])[None] + sV[None] * (np.random.rand(100, 3) - 0.5)
    assert np.array_equal(out, np.concatenate([pos, np.random.rand(100)[:, None] * 1.0], -1))
    with pytest.raises(AssertionError):
        K.init_pcd(None, None, cfg, n_points=10 ** 6, rng=np.random.RandomState(0))
