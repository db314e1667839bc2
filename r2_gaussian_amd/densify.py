"""Adaptive density control on the MI355X kernels (csrc/densify_ops.hip): the densification statistics of a
rendered view and ``densify_and_prune`` of r2_gaussian/gaussian/gaussian_model.py:320-556 -- clone, split, prune and the
Adam-state surgery -- without boolean-mask indexing (every ``x[mask]`` in the reference is a device synchronisation) and with
one host read per call (the new number of Gaussians).

Works on plain tensors; ``densify_and_prune_optimizer`` applies the result to a ``torch.optim.Adam`` whose four parameter groups
are named xyz / density / scaling / rotation like the reference's (gaussian_model.py:192-215).
"""
import ctypes as C

import torch

from . import _lib
from ._C import _on_device, _require_gpu, _stream

NAMES = ("xyz", "density", "scaling", "rotation")
_F32 = torch.float32


def densification_stats(radii, viewspace_grad, max_radii2D, grad_accum, denom):
    """In place (train.py:151-154, gaussian_model.py:552-556): for radii > 0: max_radii2D = max(., radii),
    grad_accum += ||viewspace_grad[:, :2]||, denom += 1.  One launch, no host synchronisation."""
    _require_gpu(viewspace_grad, "viewspace_grad")
    P = radii.shape[0]
    assert viewspace_grad.shape == (P, 3) and viewspace_grad.is_contiguous() and radii.dtype == torch.int32
    for t in (max_radii2D, grad_accum, denom):
        assert t.numel() == P and t.is_contiguous() and t.dtype == _F32
    dev = viewspace_grad.device
    with _on_device(dev):
        rc = _lib.lib().r2_densify_stats(P, radii.contiguous().data_ptr(), viewspace_grad.data_ptr(), max_radii2D.data_ptr(),
                                         grad_accum.data_ptr(), denom.data_ptr(), _stream(dev))
    _lib.check(rc, "r2_densify_stats")


def _ptrs(ts):
    arr = (C.c_void_p * 4)()
    for i, t in enumerate(ts):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def densify_and_prune(params, moments, max_radii2D, grad_accum, denom, normals, grad_threshold, scale_threshold, density_min,
                      bbox, scale_bound=None, do_densify=True, max_screen_size=None, max_scale=None):
    """params: dict name -> raw parameter tensor ([P,3], [P,1], [P,3], [P,4]); moments: dict name -> (exp_avg, exp_avg_sq) or
    None; normals: [2,P,3] N(0,1).  -> (new_params, new_moments, new_max_radii2D, new_grad_accum, new_denom, counts) with
    rows ordered like the reference's result; counts = surviving (originals, clones, first children, second children).
    max_screen_size / max_scale: the reference's optional prune thresholds (gaussian_model.py:540-545; None / 0 = off):
    rows whose max_radii2D exceeds the first or whose largest activated scale exceeds the second are pruned -- children of a
    split inherit the parent's max_radii2D and get its scale / 1.6.  Raises ValueError when nothing survives (train.py:169-172)."""
    xyz = params["xyz"]
    _require_gpu(xyz, "xyz")
    dev, P = xyz.device, xyz.shape[0]
    L = _lib.lib()
    ps = [params[n].detach().to(_F32).contiguous() for n in NAMES]
    have_m = moments is not None and all(moments.get(n) is not None for n in NAMES)
    ms = [moments[n][0].contiguous() for n in NAMES] if have_m else [None] * 4
    vs = [moments[n][1].contiguous() for n in NAMES] if have_m else [None] * 4
    ga, dn, mr = (t.reshape(-1).to(_F32).contiguous() for t in (grad_accum, denom, max_radii2D))
    nrm = normals.to(device=dev, dtype=_F32).contiguous()
    assert nrm.shape == (2, P, 3)
    lo, hi = (float(scale_bound[0]), float(scale_bound[1])) if scale_bound is not None else (1.0, 0.0)
    box = (C.c_float * 6)(*[float(v) for v in torch.as_tensor(bbox).reshape(-1).tolist()])
    scratch = torch.empty(L.r2_densify_scratch_bytes(P), dtype=torch.uint8, device=dev)
    counts = (C.c_uint * 4)()
    common = (float(grad_threshold), float(scale_threshold), float(density_min), box, lo, hi, int(bool(do_densify)),
              float(max_screen_size or 0.0), float(max_scale or 0.0))
    with _on_device(dev):
        rc = L.r2_densify_classify(P, ps[0].data_ptr(), ps[1].data_ptr(), ps[2].data_ptr(), ps[3].data_ptr(), mr.data_ptr(),
                                   ga.data_ptr(), dn.data_ptr(), nrm.data_ptr(), *common, scratch.data_ptr(), counts, _stream(dev))
        _lib.check(rc, "r2_densify_classify")
        cnt = tuple(int(c) for c in counts)
        Pn = sum(cnt)
        if Pn == 0:
            raise ValueError("No Gaussian left. Change adaptive control hyperparameters!")   # train.py:169-172
        widths = (3, 1, 3, 4)
        po = [torch.empty((Pn, w), dtype=_F32, device=dev) for w in widths]
        mo = [torch.empty((Pn, w), dtype=_F32, device=dev) if have_m else None for w in widths]
        vo = [torch.empty((Pn, w), dtype=_F32, device=dev) if have_m else None for w in widths]
        mro, gao, dno = (torch.empty(Pn, dtype=_F32, device=dev) for _ in range(3))
        rc = L.r2_densify_emit(P, _ptrs(ps), _ptrs(ms) if have_m else None, _ptrs(vs) if have_m else None, mr.data_ptr(),
                               ga.data_ptr(), dn.data_ptr(), nrm.data_ptr(), *common, scratch.data_ptr(), _ptrs(po),
                               _ptrs(mo) if have_m else None, _ptrs(vo) if have_m else None, mro.data_ptr(), gao.data_ptr(),
                               dno.data_ptr(), _stream(dev))
        _lib.check(rc, "r2_densify_emit")
    new_params = dict(zip(NAMES, po))
    new_moments = {n: (m, v) for n, m, v in zip(NAMES, mo, vo)} if have_m else None
    return new_params, new_moments, mro, gao, dno, cnt


def densify_and_prune_optimizer(optimizer, max_radii2D, grad_accum, denom, normals, grad_threshold, scale_threshold,
                                density_min, bbox, scale_bound=None, do_densify=True, max_screen_size=None, max_scale=None):
    """The same on a ``torch.optim.Adam`` with parameter groups named xyz / density / scaling / rotation (one tensor each):
    parameters and ``exp_avg`` / ``exp_avg_sq`` are replaced like cat_tensors_to_optimizer / _prune_optimizer do
    (gaussian_model.py:335-403).  -> (dict name -> new parameter, new max_radii2D, new grad_accum [P,1], new denom [P,1])."""
    groups = {g["name"]: g for g in optimizer.param_groups}
    params = {n: groups[n]["params"][0] for n in NAMES}
    states = {n: optimizer.state.get(params[n]) for n in NAMES}
    moments = {n: (states[n]["exp_avg"], states[n]["exp_avg_sq"]) for n in NAMES} if all(
        s is not None and "exp_avg" in s for s in states.values()) else None
    new_p, new_m, mr, ga, dn, _cnt = densify_and_prune(params, moments, max_radii2D, grad_accum, denom, normals, grad_threshold,
                                                       scale_threshold, density_min, bbox, scale_bound, do_densify,
                                                       max_screen_size, max_scale)
    out = {}
    for n in NAMES:
        old = params[n]
        st = optimizer.state.pop(old, None)
        p = torch.nn.Parameter(new_p[n].requires_grad_(True))
        groups[n]["params"][0] = p
        if st is not None:
            if new_m is not None:
                st["exp_avg"], st["exp_avg_sq"] = new_m[n]
            optimizer.state[p] = st
        out[n] = p
    return out, mr, ga.reshape(-1, 1), dn.reshape(-1, 1)
