"""Timeline of ONE steady-state training iteration on the HIP kernels (train.py:97-177, 204-209) at the
headline sizes -- 512^2 detector, 32^3 TV patch, N_INIT Gaussians (default 300k), fused losses + fused densification
statistics -- and what running the TV-voxelizer branch on a second stream buys.

  phases   device time of each phase from HIP events recorded on the stream at the phase boundaries (no host synchronisation
           inside the iteration), next to the HOST time spent issuing it: where the device waits for the host, the host column
           is the larger one
  loops    iterations/s of the whole loop, without any per-iteration synchronisation (the reference synchronises every
           iteration, train.py:147, and reads three losses back with .item(), train.py:204-209):
             serial      raster fwd -> image loss -> TV query -> TV loss -> backward -> statistics -> Adam, one stream
             tv_stream   the TV branch (voxelizer forward + TV loss, and through autograd its backward) on a second stream;
                         it is independent of the raster branch until the gradients are summed
             sync_each   serial + torch.cuda.synchronize() per iteration (the reference's loop shape)

    python scripts/profile_train.py            -> gpurun_out/profile_train.json + a markdown table on stdout
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from r2_gaussian_amd import densify as FD  # noqa: E402
from r2_gaussian_amd import losses as FL  # noqa: E402
from r2_gaussian_amd import scene as S  # noqa: E402
from tests import mini_trainer as T  # noqa: E402


def build(n_init, detector):
    """Model + views + targets without the CPU oracle (a timing run needs shapes, not a meaningful target)."""
    be = T.Backend("hip")
    dev = be.device
    views = S.make_views(50, (detector, detector))
    cloud = S.make_cloud(n_init, seed=0)
    opt = T.Opt(iterations=30000)
    raw = {"xyz": cloud.xyz, "density": T.Model.inv_softplus(cloud.density),
           "scaling": torch.log((cloud.scales - 0.001) / (1.0 - 0.001) / (1 - (cloud.scales - 0.001) / (1.0 - 0.001))),
           "rotation": cloud.rotations}
    model = T.Model.from_tensors(opt, be, raw)
    gts = [torch.rand(1, detector, detector, device=dev) * 0.5 for _ in range(4)]
    return be, dev, views, model, gts, opt


class Phases:
    def __init__(self):
        self.names, self.ev, self.host = [], [], []

    def mark(self, name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.names.append(name)
        self.ev.append(e)
        self.host.append(time.perf_counter())


def iteration(it, be, dev, views, model, gts, opt, gen, tvN, tvS, bbox, ph=None, tv_stream=None):
    mark = ph.mark if ph is not None else (lambda _n: None)
    mark("start")
    model.update_lr(it)
    x, d, s, r = model.activated()
    mark("activations")
    c = (bbox[0] + tvS / 2) + (bbox[1] - tvS - bbox[0]) * torch.rand(3, generator=gen)
    vol_loss = None
    if tv_stream is not None:   # the TV branch first, on its own stream (its forward blocks the host only until num_rendered)
        tv_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(tv_stream):
            vol = be.query(x, d, s, r, c, tvN, tvS)
            vol_loss = opt.lambda_tv * FL.tv_3d_loss(vol)
    vi = it % len(views)
    pkg = be.render(views[vi], x, d, s, r)
    mark("raster_fwd")
    loss, _parts = FL.image_loss(pkg["render"], gts[it % len(gts)], opt.lambda_dssim)
    mark("image_loss")
    if tv_stream is None:
        vol = be.query(x, d, s, r, c, tvN, tvS)
        mark("tv_query_fwd")
        vol_loss = opt.lambda_tv * FL.tv_3d_loss(vol)
        mark("tv_loss")
    else:
        torch.cuda.current_stream().wait_stream(tv_stream)
    (loss + vol_loss).backward()
    mark("backward")
    with torch.no_grad():
        FD.densification_stats(pkg["radii"], pkg["viewspace_points"].grad, model.max_radii2D, model.grad_accum, model.denom)
        mark("densify_stats")
        model.optimizer.step()
        model.optimizer.zero_grad(set_to_none=True)
    mark("adam")


def main():
    n_init = int(os.environ.get("N_INIT", "300000"))
    detector = int(os.environ.get("DETECTOR", "512"))
    be, dev, views, model, gts, opt = build(n_init, detector)
    gen = torch.Generator().manual_seed(0)
    tvN = torch.tensor([32] * 3)
    tvS = torch.tensor([2.0 / 256] * 3) * tvN
    bbox = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    args = (be, dev, views, model, gts, opt, gen, tvN, tvS, bbox)
    for it in range(1, 60):
        iteration(it, *args)
    torch.cuda.synchronize()

    # ---- phases of steady-state iterations (events on the stream; host stamps beside them)
    acc_dev, acc_host, n = {}, {}, 0
    for it in range(60, 160):
        ph = Phases()
        iteration(it, *args, ph=ph)
        torch.cuda.synchronize()
        for i in range(1, len(ph.names)):
            acc_dev[ph.names[i]] = acc_dev.get(ph.names[i], 0.0) + ph.ev[i - 1].elapsed_time(ph.ev[i]) * 1e3
            acc_host[ph.names[i]] = acc_host.get(ph.names[i], 0.0) + (ph.host[i] - ph.host[i - 1]) * 1e6
        n += 1
    phases = {k: {"device_us": round(acc_dev[k] / n, 1), "host_issue_us": round(acc_host[k] / n, 1)} for k in acc_dev}

    # ---- whole loops
    def loop(nit, tv_stream=None, sync_each=False):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(1000, 1000 + nit):
            iteration(it, *args, tv_stream=tv_stream)
            if sync_each:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return nit / (time.perf_counter() - t0)
    sB = torch.cuda.Stream(device=dev)
    loops = {}
    for name, kw in (("serial", {}), ("tv_stream", {"tv_stream": sB}), ("sync_each", {"sync_each": True}), ("serial_again", {})):
        loop(50, **kw)
        loops[name] = round(max(loop(300, **kw) for _ in range(3)), 1)
    # the same loop with torch's own fused Adam (one multi-tensor kernel per group instead of ~15 element-wise launches): a
    # one-line change in GaussianModel.training_setup (gaussian_model.py:192-215); densification's optimizer surgery is unaffected
    # (same state keys)
    try:
        groups = [{"params": g["params"], "lr": g["lr"], "name": g["name"]} for g in model.optimizer.param_groups]
        model.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15, fused=True)
        loop(50)
        loops["serial_fused_adam"] = round(max(loop(300) for _ in range(3)), 1)
    except Exception as ex:   # older torch builds: no fused Adam on this device
        loops["serial_fused_adam"] = "unavailable: %s" % str(ex)[:80]
    out = {"P": n_init, "detector": detector, "phases": phases,
           "device_us_per_iteration": round(sum(v["device_us"] for v in phases.values()), 1),
           "host_issue_us_per_iteration": round(sum(v["host_issue_us"] for v in phases.values()), 1),
           "iterations_per_s": loops}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "profile_train.json"), "w"), indent=1)
    print("| phase | device us | host issue us |\n|---|---|---|")
    for k, v in phases.items():
        print("| %s | %.1f | %.1f |" % (k, v["device_us"], v["host_issue_us"]))
    print("| **sum** | %.1f | %.1f |" % (out["device_us_per_iteration"], out["host_issue_us_per_iteration"]))
    print(json.dumps(loops))


if __name__ == "__main__":
    main()
