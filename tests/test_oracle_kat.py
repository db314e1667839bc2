"""CPU known-answer tests of the oracle with closed forms (the reference's disabled one-Gaussian debug scene,
r2_gaussian/gaussian/gaussian_model.py:166-186: xyz 0, density 0.8, scale 0.5, identity quaternion)."""
import numpy as np

from r2_gaussian_amd import scene as S
from tests import helpers as Hh


def _one(scale=0.5, density=0.8):
    xyz = np.zeros((1, 3), np.float32)
    return xyz, np.full((1, 1), density, np.float32), np.full((1, 3), scale, np.float32), np.array([[1, 0, 0, 0]], np.float32)


def test_parallel_beam_line_integral(oracle):
    """Parallel beam through an isotropic Gaussian: line integral at offset d is rho*sqrt(2pi)*sigma*exp(-d^2/2sigma^2)."""
    HW = 64
    v = S.make_view(0.0, (HW, HW), S.PARALLEL_BEAM)
    xyz, rho, sc, q = _one(0.25, 0.8)
    vm, pm = Hh.np_view(v)
    st = oracle.raster_forward(xyz, rho, sc, q, 1.0, None, vm, pm, v.tanfovx, v.tanfovy, HW, HW, v.mode)
    img = st["color"][0]
    # pixel pitch: NDC [-1,1] over HW pixels -> 2/HW world units per pixel; centre sits between pixels 31 and 32
    px = (np.arange(HW) + 0.5) * 2.0 / HW - 1.0
    X, Y = np.meshgrid(px, px)
    want = 0.8 * np.sqrt(2 * np.pi) * 0.25 * np.exp(-(X ** 2 + Y ** 2) / (2 * 0.25 ** 2))
    inside = st["n_contrib"].reshape(HW, HW) > 0
    assert inside.sum() > 500
    np.testing.assert_allclose(img[inside], want[inside], rtol=2e-5)
    assert st["radii"][0] in (24, 25)   # ceil(3*sqrt(lambda)) with lambda = 64 up to float rounding
    assert abs(st["mus"][0] - np.sqrt(2 * np.pi) * 0.25) < 1e-6


def test_voxel_centre_value(oracle):
    """Voxelizer: value at a voxel centre is rho*exp(-|x|^2/(2 sigma^2)); the cut-off is alpha < 1e-6."""
    xyz, rho, sc, q = _one(0.2, 0.8)
    n = 32
    st = oracle.voxel_forward(xyz, rho, sc, q, 1.0, None, (n, n, n), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0))
    c = (np.arange(n) + 0.5) * 2.0 / n - 1.0
    X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
    want = 0.8 * np.exp(-(X ** 2 + Y ** 2 + Z ** 2) / (2 * 0.2 ** 2))
    got = st["vol"]
    hit = st["n_contrib"].reshape(n, n, n) > 0
    assert hit.sum() > 1000
    np.testing.assert_allclose(got[hit], want[hit], rtol=3e-5)
    assert (want[~hit & (got == 0)] < 1.0).all()
    assert st["radii_x"][0] in (10, 11)   # ceil(3*0.2/0.0625 = 9.6) = 10


def test_cone_beam_magnification(oracle):
    """Cone beam: a small Gaussian at the isocentre projects to the detector centre with magnification DSD/DSO."""
    HW = 128
    v = S.make_view(1.234, (HW, HW), S.CONE_BEAM)
    xyz, rho, sc, q = _one(0.05, 0.5)
    vm, pm = Hh.np_view(v)
    st = oracle.raster_forward(xyz, rho, sc, q, 1.0, None, vm, pm, v.tanfovx, v.tanfovy, HW, HW, v.mode)
    np.testing.assert_allclose(st["means2D"][0], [(HW - 1) / 2.0, (HW - 1) / 2.0], atol=1e-3)
    np.testing.assert_allclose(st["depths"][0], 5.0, rtol=1e-6)
    # sigma on the detector in pixels = sigma * (DSD/DSO) / (sDetector/HW) = sigma * focal / depth
    focal = HW / (2 * v.tanfovx)
    a = 1.0 / st["conic_opacity"][0, 0]
    np.testing.assert_allclose(np.sqrt(a), 0.05 * focal / 5.0, rtol=1e-4)
    peak = st["color"][0].max()
    # the centre falls between four pixels: the nearest pixel centre is (0.5, 0.5) px away
    spx = 0.05 * focal / 5.0
    want = 0.5 * np.sqrt(2 * np.pi) * 0.05 * np.exp(-0.5 / (2 * spx ** 2))
    assert abs(peak - want) < 2e-3 * want


def test_knn_bruteforce(oracle):
    rng = np.random.default_rng(3)
    p = rng.random((700, 3), dtype=np.float32)
    got = oracle.knn_dist2(p)
    d = ((p[:, None, :].astype(np.float64) - p[None].astype(np.float64)) ** 2).sum(-1)
    np.fill_diagonal(d, np.inf)
    want = np.sort(d, 1)[:, :3].mean(1)
    np.testing.assert_allclose(got, want, rtol=2e-6)
    assert oracle.knn_dist2(np.zeros((0, 3), np.float32)).shape == (0,)
