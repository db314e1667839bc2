"""StreamPool (r2_gaussian_amd/multistream.py): independent views rendered concurrently on worker threads / streams give the
images of the serial loop bit for bit, and the summed gradients equal the serially accumulated ones up to the association of
the final sum over views."""
import pytest
import torch

from r2_gaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, _C
from r2_gaussian_amd import scene as S
from r2_gaussian_amd.multistream import StreamPool

pytestmark = pytest.mark.gpu


def test_pool_equals_serial(gpu):
    if _C._shim() is None:
        pytest.skip("needs the compiled boundary (_r2shim.so): it releases the GIL inside the calls")
    c = S.make_cloud(20000, seed=31)
    views = S.make_views(6, (112, 128))
    dL = S.make_pixel_grad(112, 128).to(gpu)
    params = [t.to(gpu) for t in (c.xyz, c.density, c.scales, c.rotations)]
    rast = [GaussianRasterizer(GaussianRasterizationSettings(
        image_height=112, image_width=128, tanfovx=v.tanfovx, tanfovy=v.tanfovy, scale_modifier=1.0,
        viewmatrix=v.world_view_transform.to(gpu), projmatrix=v.full_proj_transform.to(gpu), campos=v.camera_center.to(gpu),
        prefiltered=False, mode=v.mode, debug=False)) for v in views]

    def one(k, leaves):
        m2d = torch.zeros(20000, 3, device=gpu, requires_grad=True)
        img, radii = rast[k](leaves[0], m2d, leaves[1], leaves[2], leaves[3])
        img.backward(dL)
        return img.detach(), radii, m2d.grad

    # serial reference: per-view gradients, summed in view order
    ser, gsum = [], None
    for k in range(len(views)):
        lv = StreamPool.leaves_like(params)
        ser.append(one(k, lv))
        g = [p.grad for p in lv]
        gsum = g if gsum is None else [a + b for a, b in zip(gsum, g)]
    torch.cuda.synchronize()
    with StreamPool(2, gpu) as pool:
        leaves = [pool.leaves_like(params) for _ in range(pool.n)]
        for _rep in range(3):
            out = pool.map(lambda k, _v: one(k, leaves[k % pool.n]), views)
            grads = pool.sum_grads(leaves)
            torch.cuda.synchronize()
            for a, b in zip(ser, out):
                assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
            for a, b in zip(gsum, grads):
                scale = float(a.abs().max())
                assert float((a - b).abs().max()) <= 1e-6 * scale   # same addends, different association of the sum over views
        with pytest.raises(ZeroDivisionError):
            pool.map(lambda k, _v: 1 // 0, [0])


def test_a_failed_job_is_raised_only_after_every_worker_has_finished(gpu):
    """map() must not hand control back while other workers still run on shared storage."""
    import time
    done = []

    def job(k, _item):
        if k == 0:
            raise ValueError("first job fails at once")
        time.sleep(0.2)
        done.append(k)
        return k

    with StreamPool(2, gpu) as pool:
        with pytest.raises(ValueError):
            pool.map(job, range(4))
        assert sorted(done) == [1, 2, 3]          # the slower jobs completed before the exception surfaced


def test_ctypes_boundary_keeps_the_state_buffers_of_concurrent_threads_apart(gpu, monkeypatch):
    """Without the compiled boundary the allocation callbacks are Python functions shared per device; with several host
    threads inside a forward at once (ctypes releases the GIL) each must fill ITS call's state."""
    monkeypatch.setattr(_C, "_SHIM", None)          # force the ctypes boundary for this test
    monkeypatch.setattr(_C, "_SHIM_TRIED", True)
    assert _C._shim() is None
    c = S.make_cloud(30000, seed=7)
    views = S.make_views(4, (96, 96))
    params = [t.to(gpu) for t in (c.xyz, c.density, c.scales, c.rotations)]
    rast = [GaussianRasterizer(GaussianRasterizationSettings(
        image_height=96, image_width=96, tanfovx=v.tanfovx, tanfovy=v.tanfovy, scale_modifier=1.0,
        viewmatrix=v.world_view_transform.to(gpu), projmatrix=v.full_proj_transform.to(gpu), campos=v.camera_center.to(gpu),
        prefiltered=False, mode=v.mode, debug=False)) for v in views]
    dL = S.make_pixel_grad(96, 96).to(gpu)

    def one(k, leaves):
        m2d = torch.zeros(30000, 3, device=gpu, requires_grad=True)
        img, _radii = rast[k](leaves[0], m2d, leaves[1], leaves[2], leaves[3])
        img.backward(dL)
        return img.detach()

    ser = [one(k, StreamPool.leaves_like(params)) for k in range(len(views))]
    torch.cuda.synchronize()
    with StreamPool(2, gpu) as pool:
        leaves = [pool.leaves_like(params) for _ in range(pool.n)]
        for _rep in range(5):
            out = pool.map(lambda k, _v: one(k, leaves[k % pool.n]), views)
            torch.cuda.synchronize()
            for a, b in zip(ser, out):
                assert torch.equal(a, b)
