"""CPU, build container only: the oracle against oracle/_ref (the reference's CUDA kernels run on the CPU) on more and
larger seeded scenes than the committed goldens hold, plus the empty / fully-culled inputs.  Skipped where neither the
prebuilt oracle/_ref libraries nor /root/reference exist."""
import numpy as np
import pytest

from r2_gaussian_amd import scene as S
from tests import helpers as Hh

ref = pytest.importorskip("oracle.ref")
pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built and no reference tree")

FW_RASTER = ("radii", "tiles_touched", "offsets", "keys_unsorted", "vals_unsorted", "keys", "point_list", "ranges",
             "n_contrib", "depths", "means2D", "cov3D", "conic_opacity", "mus", "color")
FW_VOXEL = ("radii_x", "radii_y", "radii_z", "tiles_touched", "offsets", "keys_unsorted", "vals_unsorted", "keys",
            "point_list", "ranges", "n_contrib", "depths", "means3D_norm", "cov3D", "conic_opacity", "vol")


def _same(a, b):
    return np.array_equal(a, b) or (a.dtype == np.float32 and np.array_equal(a.view(np.uint32), b.view(np.uint32)))


@pytest.mark.parametrize("P,H,W,scanner,angle,sm", [
    (5000, 64, 64, S.CONE_BEAM, 0.3, 1.0),       # BASELINE config A
    (3000, 50, 70, S.CONE_BEAM, 2.1, 1.5),
    (4000, 96, 80, S.PARALLEL_BEAM, 1.0, 1.0),
    (20000, 256, 256, S.CONE_BEAM, 5.0, 1.0),
], ids=["A_5k_64", "ragged", "parallel", "20k_256"])
def test_raster(P, H, W, scanner, angle, sm, oracle):
    c = S.make_cloud(P, seed=P % 97, scanner=scanner, scale_mult=sm)
    v = S.make_view(angle, (H, W), scanner)
    xyz, rho, sc, q = Hh.cloud_np(c)
    vm, pm = Hh.np_view(v)
    r = ref.raster_forward(xyz, rho, sc, q, 1.0, None, vm, pm, v.tanfovx, v.tanfovy, H, W, v.mode)
    o = oracle.raster_forward(xyz, rho, sc, q, 1.0, None, vm, pm, v.tanfovx, v.tanfovy, H, W, v.mode)
    assert o["num_rendered"] == r["num_rendered"] > 0
    for k in FW_RASTER:
        assert _same(o[k], r[k]), k
    dL = S.make_pixel_grad(H, W).numpy()
    gr = ref.raster_backward(r, xyz, sc, q, 1.0, None, vm, pm, v.tanfovx, v.tanfovy, dL)
    go = oracle.raster_backward(o, xyz, sc, q, 1.0, None, vm, pm, v.tanfovx, v.tanfovy, dL, acc64=False)
    for k in gr:
        scale = max(float(np.abs(gr[k]).max()), 1e-30)
        assert float(np.abs(go[k] - gr[k]).max()) <= 2e-5 * scale, k


@pytest.mark.parametrize("P,n,sv,ctr", [(3000, (32, 32, 32), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0)),
                                         (4000, (24, 40, 17), (1.5, 2.5, 1.0625), (0.1, -0.2, 0.05)),
                                         (8000, (32, 32, 32), (0.25, 0.25, 0.25), (0.3, 0.1, -0.2))],
                         ids=["32cube", "ragged", "tv_subvolume"])
def test_voxel(P, n, sv, ctr, oracle):
    c = S.make_cloud(P, seed=P % 89)
    xyz, rho, sc, q = Hh.cloud_np(c)
    r = ref.voxel_forward(xyz, rho, sc, q, 1.0, None, n, sv, ctr)
    o = oracle.voxel_forward(xyz, rho, sc, q, 1.0, None, n, sv, ctr)
    assert o["num_rendered"] == r["num_rendered"] > 0
    for k in FW_VOXEL:
        assert _same(o[k], r[k]), k
    rng = np.random.default_rng(1)
    dL = ((rng.random(n, dtype=np.float32) * 2 - 1) / np.prod(n)).astype(np.float32)
    gr = ref.voxel_backward(r, xyz, sc, q, 1.0, None, dL)
    go = oracle.voxel_backward(o, sc, q, 1.0, None, dL, acc64=False)
    for k in gr:
        scale = max(float(np.abs(gr[k]).max()), 1e-30)
        assert float(np.abs(go[k] - gr[k]).max()) <= 2e-5 * scale, k


def test_empty_and_culled(oracle):
    v = S.make_view(0.4, (32, 32))
    vm, pm = Hh.np_view(v)
    e3, e1, e4 = np.zeros((0, 3), np.float32), np.zeros((0, 1), np.float32), np.zeros((0, 4), np.float32)
    r = ref.raster_forward(e3, e1, e3, e4, 1.0, None, vm, pm, v.tanfovx, v.tanfovy, 32, 32, v.mode)
    o = oracle.raster_forward(e3, e1, e3, e4, 1.0, None, vm, pm, v.tanfovx, v.tanfovy, 32, 32, v.mode)
    assert r["num_rendered"] == o["num_rendered"] == 0 and not r["color"].any() and not o["color"].any()
    # every Gaussian behind the source: num_rendered == 0, image all zero, all radii zero
    c = S.make_cloud(50, seed=1)
    xyz = c.xyz.numpy() * 0.01 + np.array([6.0 * np.cos(0.4), 6.0 * np.sin(0.4), 0.0], np.float32)
    rho, sc, q = c.density.numpy(), c.scales.numpy(), c.rotations.numpy()
    r = ref.raster_forward(xyz, rho, sc, q, 1.0, None, vm, pm, v.tanfovx, v.tanfovy, 32, 32, v.mode)
    o = oracle.raster_forward(xyz, rho, sc, q, 1.0, None, vm, pm, v.tanfovx, v.tanfovy, 32, 32, v.mode)
    assert r["num_rendered"] == o["num_rendered"] == 0
    assert not r["radii"].any() and not o["radii"].any() and not r["color"].any() and not o["color"].any()
    assert np.array_equal(ref.mark_visible(xyz, vm, pm), oracle.mark_visible(xyz, vm, pm))
    assert not oracle.mark_visible(xyz, vm, pm).any()


def test_mark_visible(oracle):
    c = S.make_cloud(500, seed=2)
    v = S.make_view(1.0, (32, 32))
    xyz = c.xyz.numpy() * 9.0   # some behind the source (DSO = 5)
    vm, pm = Hh.np_view(v)
    a, b = ref.mark_visible(xyz, vm, pm), oracle.mark_visible(xyz, vm, pm)
    assert np.array_equal(a, b) and a.any() and not a.all()
