"""Two host threads drive the library at the same time, each on its own stream (the compiled torch boundary releases the GIL
inside the calls).  The library's mutable state -- depth-range history, read-back mailbox, error text -- is per host thread;
nothing may leak between the threads: every result equals the one computed serially, bit for bit."""
import threading

import pytest
import torch

from r2_gaussian_amd import _C
from r2_gaussian_amd import scene as S

pytestmark = pytest.mark.gpu


def _work(c, views, dev, dL, n_vox):
    out = []
    e = torch.empty(0)
    for v in views:
        a = (c.xyz.to(dev), c.density.to(dev), c.scales.to(dev), c.rotations.to(dev), 1.0, e, v.world_view_transform.to(dev),
             v.full_proj_transform.to(dev), v.tanfovx, v.tanfovy, v.image_height, v.image_width, v.camera_center.to(dev), False,
             v.mode, False)
        R, col, rad, g, b, i = _C.rasterize_gaussians(*a)
        gr = _C.rasterize_gaussians_backward(a[0], rad, a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], dL, a[12], g, R, b, i,
                                             v.mode, False)
        Rv, vol, rx, ry, rz, gv, bv, iv = _C.voxelize_gaussians(a[0], a[1], a[2], a[3], 1.0, e, n_vox, n_vox, n_vox, 2.0, 2.0, 2.0,
                                                                0.0, 0.0, 0.0, False, False)
        out.append((R, col.clone(), rad.clone(), [t.clone() for t in gr], Rv, vol.clone()))
    return out


def test_two_threads_two_streams(gpu):
    if _C._shim() is None:
        pytest.skip("_r2shim.so not built: the ctypes boundary shares its allocation hooks per device (documented)")
    jobs = [(S.make_cloud(30000, seed=21), S.make_views(12, (128, 144)), S.make_pixel_grad(128, 144).to(gpu), 40),
            (S.make_cloud(11000, seed=22), S.make_views(12, (96, 80)), S.make_pixel_grad(96, 80).to(gpu), 24)]
    serial = [_work(c, v, gpu, dL, n) for c, v, dL, n in jobs]
    torch.cuda.synchronize()
    results, errors = [None, None], []

    def run(k):
        try:
            st = torch.cuda.Stream(device=gpu)
            with torch.cuda.stream(st):
                for _ in range(3):
                    results[k] = _work(*jobs[k][:2], gpu, jobs[k][2], jobs[k][3])
                st.synchronize()
        except Exception as ex:   # noqa: BLE001
            errors.append(ex)

    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    torch.cuda.synchronize()
    for k in range(2):
        for a, b in zip(serial[k], results[k]):
            assert a[0] == b[0] and a[4] == b[4]
            assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[5], b[5])
            for x, y in zip(a[3], b[3]):
                assert torch.equal(x, y)
