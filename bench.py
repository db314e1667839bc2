#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native R2-Gaussian hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Metric (BASELINE.json): rasterized X-ray views/s (forward + backward) at 300k Gaussians on a 512^2 cone-beam
detector, plus voxelizer GVoxel/s at 300k Gaussians / 256^3 (reported in the same JSON line).

A "step" = one training view through the drop-in surface: GaussianRasterizer forward (autograd) + backward
with a fixed upstream gradient dL/dpix, i.e. r2_raster_forward + r2_raster_backward of the C ABI, inputs
resident in HBM.  With N > 1 ranks (launched by torch.distributed.run, one per GPU) every rank renders its
own view of the 50-view training set and the packed [P,11] parameter gradients are all-reduced over
RCCL/xGMI each step (weak scaling: per-GPU work fixed).  value = all views of all ranks / max-over-ranks time.

The JSON line also carries
  roofline     -- dominant kernel: algorithmic bytes per launch / HIP-event duration vs the 8 TB/s HBM peak
  cpu_baseline -- the CPU oracle (a port of the reference's algorithm, oracle/r2_oracle.c) timed on the host
                  cores for ONE view of the same workload (rank 0, N == 1 only)
  kernels      -- per-stage HIP-event breakdown from a second, instrumented pass (not part of `value`)
Synthetic seeded data (no datasets offline), random Gaussian cloud of the named size.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def algorithmic_bytes(stage, P, R, T, N):
    """Algorithmic HBM bytes per launch (DESIGN.md 5 / SURVEY.md 8d): every array counted once read + once
    written, the sort as ONE read + write of (key, value)."""
    table = {
        "raster.preprocess": 108 * P,            # 44 in + 64 out
        "raster.depth_sort": 16 * P,             # depth order of the Gaussians: (key, id) read + id written
        "raster.scan": 8 * P,
        "raster.duplicate": 20 * P + 12 * R,
        "raster.sort": 24 * R,
        "raster.ranges": 8 * R + 8 * T,
        "raster.render_fwd": 32 * R + 8 * T + 8 * N,
        "raster.render_bwd": 32 * R + 8 * N + 28 * P,
        "raster.geom_bwd": 228 * P,              # cov2D 96 + preprocess 132
    }
    return table.get(stage, 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--gaussians", type=int, default=300000)
    ap.add_argument("--detector", type=int, default=512)
    ap.add_argument("--views", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-voxel", action="store_true")
    args = ap.parse_args()

    import torch.distributed as dist
    from r2_gaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib
    from r2_gaussian_amd import dist as r2dist
    from r2_gaussian_amd import scene as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    all_cpus = os.sched_getaffinity(0)
    pinned = r2dist.pin_to_gpu_numa_node(local_rank)   # one process per GPU, on a slice of the GPU's own socket
    if world > 1 or os.environ.get("R2_BENCH_FORCE_COMM", "0") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    _lib.lib()

    P, HW = args.gaussians, args.detector
    cloud = S.make_cloud(P, seed=0)
    views = S.make_views(args.views, (HW, HW))
    xyz = cloud.xyz.to(dev).requires_grad_(True)
    dens = cloud.density.to(dev).requires_grad_(True)
    scal = cloud.scales.to(dev).requires_grad_(True)
    rot = cloud.rotations.to(dev).requires_grad_(True)
    params = (xyz, dens, scal, rot)
    dL = S.make_pixel_grad(HW, HW).to(dev)
    settings = [GaussianRasterizationSettings(
        image_height=HW, image_width=HW, tanfovx=v.tanfovx, tanfovy=v.tanfovy, scale_modifier=1.0,
        viewmatrix=v.world_view_transform.to(dev), projmatrix=v.full_proj_transform.to(dev),
        campos=v.camera_center.to(dev), prefiltered=False, mode=v.mode, debug=False) for v in views]
    rasterizers = [GaussianRasterizer(s) for s in settings]
    # N > 1: the packed [P,11] gradients of step k are all-reduced (RCCL, its own stream) WHILE step k+1 renders: two
    # flat buffers, each waited for before it is packed again and all of them before the clock stops.  Every step's
    # reduction is complete inside the timed region; a trainer consumes it one step late (pipelined data parallelism) or
    # accumulates several views per optimiser step.
    force_comm = os.environ.get("R2_BENCH_FORCE_COMM", "0") == "1"   # exercise the collective path on one GPU (tests)
    use_comm = world > 1 or (force_comm and dist.is_initialized())
    flats = [torch.empty((P, r2dist.GRAD_WIDTH), dtype=torch.float32, device=dev) for _ in range(2)]
    pending = [None, None]
    stats = {"R": 0}

    # the screen-space placeholder whose only role is to receive dL/dmeans2D (render_query.py:113-120): an input like
    # the parameters, resident before the timed region
    means2D = torch.zeros_like(xyz, requires_grad=True)

    def step(k):
        vi = r2dist.view_for(k, len(views), rank_=rank, world_=world)
        img, radii = rasterizers[vi](means3D=xyz, means2D=means2D, opacities=dens, scales=scal, rotations=rot)
        means2D.grad = None
        for p in params:
            p.grad = None
        img.backward(dL)
        stats["R"] = img.grad_fn.num_rendered if hasattr(img.grad_fn, "num_rendered") else stats["R"]
        if use_comm:
            i = k & 1
            if pending[i] is not None:
                pending[i][0].wait()       # the reduction started two steps ago is done
            # the backward leaves the four parameter gradients adjacent in one buffer: reduce them where they are
            blk = r2dist.grad_block(xyz.grad, dens.grad, scal.grad, rot.grad)
            if blk is None:
                blk = r2dist.pack_grads(xyz.grad, dens.grad, scal.grad, rot.grad, out=flats[i])
            stats["zero_copy"] = blk is not flats[i]
            pending[i] = (r2dist.allreduce_grads(blk, average=False, async_op=True), blk)   # keep the buffer alive
        return img

    def drain():
        for i in range(2):
            if pending[i] is not None:
                pending[i][0].wait()
                pending[i] = None

    def barrier():
        drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if os.environ.get("R2_BENCH_NOGC", "0") == "1":
        import gc
        gc.disable()
    for k in range(args.warmup):
        step(k)
    # the dominant kernel is bracketed with HIP events on its own stream inside the timed region
    DOMINANT = "raster.render_bwd"
    _lib.profile_read(reset=True)
    _lib.profile_enable([DOMINANT])
    barrier()
    _lib.sync_wait_stats(reset=True)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    barrier()
    dt = time.perf_counter() - t0
    wait_us, wait_n = _lib.sync_wait_stats(reset=True)
    dom = _lib.profile_read(reset=True).get(DOMINANT, (0.0, 0))
    _lib.profile_enable([])
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- second, instrumented pass: per-stage breakdown + R of the measured views (not part of `value`)
    _lib.profile_enable(None)
    Rs = []
    for k in range(min(args.steps, 50)):
        img = step(args.warmup + k)
        Rs.append(stats["R"])
    torch.cuda.synchronize()
    prof = _lib.profile_read(reset=True)
    _lib.profile_enable([])
    # num_rendered of the measured views via the C mirror (the autograd ctx is gone by now)
    from r2_gaussian_amd import _C
    e = torch.empty(0)
    Rl = []
    with torch.no_grad():
        for vi in range(0, len(views), max(1, len(views) // 10)):
            s = settings[vi]
            Rl.append(_C.rasterize_gaussians(xyz, dens, scal, rot, 1.0, e, s.viewmatrix, s.projmatrix, s.tanfovx,
                                             s.tanfovy, HW, HW, s.campos, False, s.mode, False)[0])
    R = int(sum(Rl) / len(Rl))
    N, T = HW * HW, ((HW + 15) // 16) ** 2
    kernels = {}
    for name, (ms, cnt) in sorted(prof.items()):
        us = 1e3 * ms / cnt
        b = algorithmic_bytes(name, P, R, T, N)
        kernels[name] = {"us": round(us, 2), "alg_MB": round(b / 1e6, 2),
                         "GBps": round(b / (us * 1e-6) / 1e9, 1) if us > 0 else None}
    dom_us = 1e3 * dom[0] / max(dom[1], 1)
    dom_bytes = algorithmic_bytes(DOMINANT, P, R, T, N)
    achieved = dom_bytes / (dom_us * 1e-6) / 1e9 if dom_us > 0 else 0.0
    total_bytes = 492 * P + 108 * R + 24 * T + 16 * N

    # ---- voxelizer GVoxel/s (second half of the BASELINE metric): 300k Gaussians on the 256^3 volume
    gvox = None
    if not args.no_voxel and rank == 0:
        with torch.no_grad():
            va = (xyz, dens, scal, rot, 1.0, e, 256, 256, 256, 2.0, 2.0, 2.0, 0.0, 0.0, 0.0, False, False)
            for _ in range(3):
                R3 = _C.voxelize_gaussians(*va)[0]
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            nv = 10
            for _ in range(nv):
                _C.voxelize_gaussians(*va)
            torch.cuda.synchronize()
            tv = (time.perf_counter() - t1) / nv
            _lib.profile_enable(None)       # per-stage breakdown of the same call (not part of the timing above)
            for _ in range(5):
                _C.voxelize_gaussians(*va)
            torch.cuda.synchronize()
            vprof = _lib.profile_read(reset=True)
            _lib.profile_enable([])
        # the training loop's TV regulariser: forward + backward of a 32^3 patch through the drop-in voxelizer
        from r2_gaussian_amd import GaussianVoxelizationSettings, GaussianVoxelizer
        vox32 = [GaussianVoxelizer(GaussianVoxelizationSettings(1.0, 32, 32, 32, 0.25, 0.25, 0.25, -0.3 + 0.1 * (i % 7),
                                                                0.1 * (i % 5) - 0.2, 0.05 * (i % 9) - 0.2, False, False))
                 for i in range(16)]
        gvol = torch.full((32, 32, 32), 1.0 / 32 ** 3, device=dev)

        def tv_step(i):
            vol, _r = vox32[i % 16](means3D=xyz, opacities=dens, scales=scal, rotations=rot)
            for p_ in params:
                p_.grad = None
            vol.backward(gvol)
        for i in range(10):
            tv_step(i)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for i in range(100):
            tv_step(i)
        torch.cuda.synchronize()
        ttv = (time.perf_counter() - t2) / 100
        vbytes = 168 * P + 88 * R3 + 16 * 32768 + 8 * 256 ** 3
        gvox = {"gvoxel_per_s": round(256 ** 3 / tv / 1e9, 3), "ms": round(tv * 1e3, 3), "R3": int(R3),
                "alg_MB": round(vbytes / 1e6, 1), "hbm_frac": round(vbytes / tv / 1e9 / HBM_PEAK_GBS, 4),
                "stages_us": {k: round(1e3 * ms / cnt, 1) for k, (ms, cnt) in sorted(vprof.items()) if k.startswith("voxel.")},
                "tv_patch_32cube_fwd_bwd_us": round(ttv * 1e6, 1)}

    # ---- CPU baseline: the oracle (port of the reference algorithm) on the host cores, one view
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        os.sched_setaffinity(0, all_cpus)   # the CPU baseline may use every host core again
        from oracle import oracle as O
        O.lib()
        v = views[0]
        xn, dn, sn, qn = (t.detach().cpu().numpy() for t in (xyz, dens, scal, rot))
        vm, pm = v.world_view_transform.numpy(), v.full_proj_transform.numpy()
        tc = time.perf_counter()
        st = O.raster_forward(xn, dn, sn, qn, 1.0, None, vm, pm, v.tanfovx, v.tanfovy, HW, HW, v.mode)
        O.raster_backward(st, xn, sn, qn, 1.0, None, vm, pm, v.tanfovx, v.tanfovy, dL.cpu().numpy(), acc64=False)
        tc = time.perf_counter() - tc
        cpu = {"value": round(1.0 / tc, 4), "unit": "views/s", "cores": int(O.lib().r2o_num_threads()), "kind": "port",
               "sample": "1 view fwd+bwd of the same workload (%dk Gaussians, %d^2, R=%d) by oracle/r2_oracle.c, OpenMP"
                         % (P // 1000, HW, st["num_rendered"])}

    # HBM traffic of the dominant kernel from the TCC counters (FETCH_SIZE / WRITE_SIZE, one counter per rocprofv3 pass:
    # scripts/gpu_pmc.sh; the summary it writes is committed under profiles/).  bench.py cannot collect PMCs itself.
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path))
            k = pmc.get("r2::raster_render_backward_kernel", {})
            if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
                traffic = int((k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024)   # counters are in KB, per launch
        except Exception:
            traffic = None

    if rank == 0:
        total_views = args.steps * world
        out = {
            "metric": "rasterized X-ray views/sec (fwd+bwd) at 300k Gaussians, 512^2 cone-beam detector",
            "value": round(total_views / dt, 2), "unit": "views/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "synthetic 0_chest_cone-like cone-beam set: %d Gaussians (seed 0), %dx%d detector, "
                                   "%d views, DSD 7 / DSO 5" % (P, HW, HW, args.views),
                       "num_rendered": R, "parallelism": "view-sharded dp%d + RCCL all-reduce of [P,11] grads (%s, overlapped with the next view)" % (
                           world, "in place, zero-copy" if stats.get("zero_copy") else "packed copy")
                       if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": DOMINANT, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "us_per_launch": round(dom_us, 2), "alg_bytes_per_launch": dom_bytes,
                         "pipeline_frac": round(total_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                         "pipeline_alg_bytes": total_bytes},
            "cpu_baseline": cpu,
            "comm_zero_copy": stats.get("zero_copy") if use_comm else None,
            # host time per step spent waiting for num_rendered at the forward's sync: large = GPU-bound step
            "host_wait_us_per_step": round(wait_us / max(wait_n, 1), 1),
            "host_cpus_pinned": len(pinned) if pinned else None,
            "kernels": kernels,
            "voxelizer": gvox,
        }
        print(json.dumps(out))
    if dist.is_initialized():
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
