"""Batched views (r2_raster_forward_batch / r2_raster_backward_batch, new functionality): V views of the same Gaussians
through ONE pass of the pipeline.  Every view must come out exactly as the single-view call renders it -- image and radii
BIT-identical (the records keep their per-view coordinates, the per-tile lists are the same lists) -- the per-view
screen-space gradients too, and the parameter gradients must be the sum over the views (checked against the oracle's
per-view double sums with the attributed-flip accounting, tolerances added over the views)."""
import numpy as np
import pytest
import torch

from r2_gaussian_amd import scene as S
from tests import helpers as Hh

pytestmark = pytest.mark.gpu


def _batch(c, views, dev, debug=False):
    from r2_gaussian_amd import _C
    e = torch.empty(0)
    vm = torch.stack([v.world_view_transform for v in views]).to(dev)
    pm = torch.stack([v.full_proj_transform for v in views]).to(dev)
    v0 = views[0]
    args = (c.xyz.to(dev), c.density.to(dev), c.scales.to(dev), c.rotations.to(dev), 1.0, e, vm, pm, v0.tanfovx, v0.tanfovy,
            v0.image_height, v0.image_width, v0.mode, debug)
    return args, _C.rasterize_gaussians_batch(*args)


@pytest.mark.parametrize("P,hw,V", [(6000, (80, 96), 3), (3000, (50, 70), 4), (20000, (256, 256), 4), (4000, (64, 64), 9),
                                    (300000, (512, 512), 4)],
                         ids=["3x80x96", "4x50x70_ragged", "4x256x256", "9x64x64_multipass_sort", "4x512x512_300k_as_benchmarked"])
def test_batch_equals_single_views_and_oracle(P, hw, V, oracle, gpu):
    _batch_case(P, hw, V, S.CONE_BEAM, oracle, gpu)


def test_batch_parallel_beam(oracle, gpu):
    _batch_case(5000, (72, 88), 3, S.PARALLEL_BEAM, oracle, gpu)


def _batch_case(P, hw, V, scanner, oracle, gpu):
    from r2_gaussian_amd import _C
    c = S.make_cloud(P, seed=P % 101)
    views = [S.make_view(0.3 + 0.9 * k, hw, scanner) for k in range(V)]
    args, (R, color, radii, gb, bb, ib) = _batch(c, views, gpu)
    torch.cuda.synchronize()
    assert color.shape == (V,) + hw and radii.shape == (V, P)
    singles = [Hh.hip_raster(c, v, gpu) for v in views]
    assert R == sum(h["num_rendered"] for h in singles)
    for k, h in enumerate(singles):
        assert np.array_equal(radii[k].cpu().numpy(), h["radii"]), "radii of view %d" % k
        assert np.array_equal(color[k].cpu().numpy().view(np.uint32), h["color"][0].view(np.uint32)), "image of view %d is not bit-identical" % k
    # backward: one upstream gradient per view
    g = torch.Generator().manual_seed(3)
    dL = ((torch.rand((V,) + hw, generator=g) * 2 - 1) / float(hw[0] * hw[1])).to(gpu)
    res = _C.rasterize_gaussians_backward_batch(args[0], radii, args[2], args[3], 1.0, args[5], args[6], args[7], args[8],
                                                args[9], dL, gb, R, bb, ib, args[12], False)
    torch.cuda.synchronize()
    names = ["dL_dmeans2D", "dL_dopacity", "dL_dmu", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations"]
    gh = {n: t.cpu().numpy() for n, t in zip(names, res)}
    assert gh["dL_dmeans2D"].shape == (V, P, 3) and gh["dL_dmu"].shape == (V, P)
    # per view: the screen-space gradients are those of the single-view backward, bit for bit; the parameter gradients sum up
    tot = {n: 0.0 for n in ("dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations")}
    for k, (v, h) in enumerate(zip(views, singles)):
        gs = Hh.hip_raster_backward(h, c, v, dL[k:k + 1].cpu().numpy(), gpu)
        assert np.array_equal(gh["dL_dmeans2D"][k].view(np.uint32), gs["dL_dmeans2D"].view(np.uint32)), "dL_dmeans2D of view %d" % k
        assert np.array_equal(gh["dL_dmu"][k].view(np.uint32), gs["dL_dmu"].reshape(-1).view(np.uint32))
        for n in tot:
            tot[n] = tot[n] + gs[n].astype(np.float64)
    for n in tot:
        scale = max(float(np.abs(tot[n]).max()), 1e-30)
        assert float(np.abs(gh[n] - tot[n]).max()) <= 2e-6 * scale, n   # same terms, summed in one kernel instead of V
    # and against the oracle, view by view (image) -- the gradients of view 0 of a V = 1 batch are covered by test_v1_is_the_reference_call
    for k, v in enumerate(views[:2]):
        o = Hh.oracle_raster(oracle, c, v)
        Hh.parity_image(oracle, o, color[k:k + 1].cpu().numpy(), "batch V=%d view %d" % (V, k))


def test_v1_is_the_reference_call(oracle, gpu):
    """A batch of one view is the single-view call: same num_rendered, bit-identical image, gradients within the oracle bound."""
    from r2_gaussian_amd import _C
    c = S.make_cloud(5000, seed=8)
    v = S.make_view(1.3, (96, 80))
    args, (R, color, radii, gb, bb, ib) = _batch(c, [v], gpu)
    h = Hh.hip_raster(c, v, gpu)
    assert R == h["num_rendered"] and np.array_equal(color.cpu().numpy().view(np.uint32), h["color"].view(np.uint32))
    dL = S.make_pixel_grad(96, 80).to(gpu)
    res = _C.rasterize_gaussians_backward_batch(args[0], radii, args[2], args[3], 1.0, args[5], args[6], args[7], args[8],
                                                args[9], dL, gb, R, bb, ib, args[12], False)
    torch.cuda.synchronize()
    names = ["dL_dmeans2D", "dL_dopacity", "dL_dmu", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations"]
    gh = {n: t.cpu().numpy() for n, t in zip(names, res)}
    gh["dL_dmeans2D"], gh["dL_dmu"] = gh["dL_dmeans2D"][0], gh["dL_dmu"][0]
    o = Hh.oracle_raster(oracle, c, v)
    Hh.parity_raster_grads(oracle, o, gh, c, v, dL.cpu().numpy(), "batch V=1")


def test_autograd_module_sums_the_views(gpu):
    from r2_gaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, GaussianRasterizerBatch
    c = S.make_cloud(3000, seed=4)
    views = [S.make_view(0.5 * k, (64, 64)) for k in range(4)]
    p = [t.to(gpu).requires_grad_(True) for t in (c.xyz, c.density, c.scales, c.rotations)]
    v0 = views[0]
    rsb = GaussianRasterizationSettings(64, 64, v0.tanfovx, v0.tanfovy, 1.0,
                                        torch.stack([v.world_view_transform for v in views]).to(gpu),
                                        torch.stack([v.full_proj_transform for v in views]).to(gpu),
                                        torch.stack([v.camera_center for v in views]).to(gpu), False, v0.mode, False)
    m2 = torch.zeros((4,) + tuple(p[0].shape), device=gpu, requires_grad=True)
    img, radii = GaussianRasterizerBatch(rsb)(means3D=p[0], means2D=m2, opacities=p[1], scales=p[2], rotations=p[3])
    w = torch.linspace(0.5, 2.0, 4, device=gpu)[:, None, None]
    (img * w).sum().backward()
    got = [t.grad.clone() for t in p] + [m2.grad.clone()]
    for t in p:
        t.grad = None
    want2 = []
    for k, v in enumerate(views):
        rs = GaussianRasterizationSettings(64, 64, v.tanfovx, v.tanfovy, 1.0, v.world_view_transform.to(gpu),
                                           v.full_proj_transform.to(gpu), v.camera_center.to(gpu), False, v.mode, False)
        m = torch.zeros_like(p[0], requires_grad=True)
        im, _r = GaussianRasterizer(rs)(means3D=p[0], means2D=m, opacities=p[1], scales=p[2], rotations=p[3])
        assert torch.equal(im[0], img[k].detach())
        (im * w[k]).sum().backward()
        want2.append(m.grad)
    for a, t in zip(got[:4], p):
        assert torch.allclose(a, t.grad, rtol=1e-5, atol=1e-6 * float(t.grad.abs().max()))
    assert torch.equal(got[4], torch.stack(want2))


def _tf_taken():
    import ctypes as C
    from r2_gaussian_amd import _lib
    st = (C.c_longlong * 5)()
    _lib.lib().r2_tile_first_stats(st, 0)
    return int(st[0])


@pytest.mark.parametrize("tile_first", [True, False], ids=["tile_first_chain", "hinted_general_chain"])
def test_second_batched_call_of_a_size(tile_first, gpu):
    """The SECOND batched call of a size runs the tile-first chain over the stacked tile grids of the views (round 6: V * T <= 4096
    tiles; the three one-round binning kernels are paid once per batch) -- or, with that chain switched off, the hinted depth order
    (sorted records, emission by output range).  Either way images, radii and per-view screen-space gradients stay bit-identical
    to single-view calls."""
    from r2_gaussian_amd import _C, _lib
    P, hw, V = 20000, (256, 256), 4
    c = S.make_cloud(P, seed=7)
    L = _lib.lib()
    L.r2_tile_first_control(1 if tile_first else 0)
    try:
        _batch(c, [S.make_view(0.1 + 0.7 * k, hw) for k in range(V)], gpu)          # first call of this size: general chain
        views = [S.make_view(0.4 + 0.8 * k, hw) for k in range(V)]
        before = _tf_taken()
        args, (R, color, radii, gb, bb, ib) = _batch(c, views, gpu)                  # predicted / hinted
        torch.cuda.synchronize()
        assert (_tf_taken() - before == 1) == tile_first, "the batched call did not take the chain it was meant to"
        singles = [Hh.hip_raster(c, v, gpu) for v in views]
        assert R == sum(h["num_rendered"] for h in singles)
        g = torch.Generator().manual_seed(5)
        dL = ((torch.rand((V,) + hw, generator=g) * 2 - 1) / float(hw[0] * hw[1])).to(gpu)
        res = _C.rasterize_gaussians_backward_batch(args[0], radii, args[2], args[3], 1.0, args[5], args[6], args[7], args[8],
                                                    args[9], dL, gb, R, bb, ib, args[12], False)
        torch.cuda.synchronize()
        d2 = res[0].cpu().numpy()
        # the stacked point_list is the views' lists one after the other (ids = view * P + Gaussian)
        import ctypes as C
        bid = C.c_int(-1)
        off = L.r2_raster_state_offset(5, P * V, R, hw[1], hw[0] * V, C.byref(bid))
        pl = bb.cpu().numpy()[off:off + 4 * R].view(np.uint32)
        want = np.concatenate([h["point_list"] + np.uint32(k * P) for k, h in enumerate(singles)])
        assert np.array_equal(pl, want), "stacked point_list differs from the single views' lists"
        for k, (v, h) in enumerate(zip(views, singles)):
            assert np.array_equal(radii[k].cpu().numpy(), h["radii"])
            assert np.array_equal(color[k].cpu().numpy().view(np.uint32), h["color"][0].view(np.uint32)), "image of view %d" % k
            gs = Hh.hip_raster_backward(h, c, v, dL[k:k + 1].cpu().numpy(), gpu)
            assert np.array_equal(d2[k].view(np.uint32), gs["dL_dmeans2D"].view(np.uint32)), "dL_dmeans2D of view %d" % k
    finally:
        L.r2_tile_first_control(1)


def test_tile_first_batch_at_the_benchmarked_size(gpu):
    """4 views of 300k Gaussians at 512^2 = 4096 stacked tiles, 1.2 M view instances: the largest batch the chain takes."""
    P, hw, V = 300000, (512, 512), 4
    c = S.make_cloud(P, seed=0)
    _batch(c, [S.make_view(0.2 + 0.5 * k, hw) for k in range(V)], gpu)
    views = [S.make_view(0.45 + 0.5 * k, hw) for k in range(V)]
    before = _tf_taken()
    args, (R, color, radii, gb, bb, ib) = _batch(c, views, gpu)
    torch.cuda.synchronize()
    assert _tf_taken() - before == 1
    for k, v in enumerate(views):
        h = Hh.hip_raster(c, v, gpu)
        assert np.array_equal(radii[k].cpu().numpy(), h["radii"])
        assert np.array_equal(color[k].cpu().numpy().view(np.uint32), h["color"][0].view(np.uint32)), "image of view %d" % k
