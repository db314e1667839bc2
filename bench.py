#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native R2-Gaussian hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--repeats M]

Metric (BASELINE.json): rasterized X-ray views/s (forward + backward) at 300k Gaussians on a 512^2 cone-beam
detector, plus voxelizer GVoxel/s at 300k Gaussians / 256^3 (reported in the same JSON line).

A "step" = one training view through the drop-in surface: GaussianRasterizer forward (autograd) + backward with a
fixed upstream gradient dL/dpix, i.e. r2_raster_forward + r2_raster_backward of the C ABI, inputs resident in HBM.

--gpus N > 1: one process per GPU.  When WORLD_SIZE is not set the script launches its own N ranks (re-exec through
torch.distributed.run on 127.0.0.1); under a launcher it checks that the world it finds IS N.  Every rank renders its own
view of the 50-view training set and the [P,11] parameter gradients are all-reduced over RCCL/xGMI each step (weak scaling:
per-GPU work fixed).  `value` is the SYNCHRONOUS mode: the all-reduce of step k is ordered before step k+1 on the stream
(what a trainer that applies every step's gradients needs, and what a PSNR study would be run with); `overlapped` in the
same line is the pipelined mode (the reduction of step k runs while step k+1 renders; gradients arrive one step late).

Timing: W untimed warm-up steps, then M (--repeats) regions of EXACTLY K steps, each bracketed by barrier +
torch.cuda.synchronize() on both sides and reduced with MAX over ranks; `value` = all views of all ranks / the MEDIAN
region (min / max / spread are in `timing`): a 20-step driver run is then 25 samples, not one 5 ms sample.

The JSON line also carries
  roofline       dominant kernel (picked from an instrumented pre-pass): algorithmic bytes per launch / its HIP-event duration
                 measured live inside the timed regions, vs the 8 TB/s HBM peak; `traffic` = FETCH_SIZE + WRITE_SIZE of that
                 kernel from the committed PMC summary, accepted only if it was collected on the same kernel sources
  cpu_baseline   the CPU oracle (C/OpenMP port of the reference algorithm, oracle/r2_oracle.c) timed on the host cores for
                 ONE view of the same workload (rank 0, N == 1 only)
  parity_checked the SAME view's GPU image and gradients checked against that oracle result (pure 1e-4 relative bound +
                 attributed cut-off flips, oracle/parity.py); the run FAILS when it is out of tolerance
  forward_only   views/s of the forward alone
  batched        views/s with four views per call (r2_raster_forward_batch / _backward_batch)
  concurrent_streams  views/s with two independent views in flight: two host threads, one HIP stream each, through the same
                 drop-in classes (the binning chain of one view overlaps the render kernels of the other)
  kernels        per-stage HIP-event breakdown + achieved fraction of the HBM peak, from an instrumented pass (not in `value`)
Synthetic seeded data (no datasets offline), random Gaussian cloud of the named size.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
VALU_SIMDS, VALU_CYCLES_PER_INST, VALU_CLOCK_GHZ = 1024, 2.8, 2.4   # 256 CUs x 4; profiles/r06_ubench_valu.txt

# stage (r2_profile_* name) -> kernel name in the rocprofv3 / PMC summaries
STAGE_KERNEL = {
    "raster.render_bwd": "r2::raster_render_backward_kernel<false>",
    "raster.render_fwd": "r2::raster_render_forward_wave_kernel<false>",   # round 6: the one-wave kernel
    "raster.geom_bwd": "r2::raster_geom_backward_kernel<false>",
    # (tile-first binning chain, rounds 4-5; the general chain's kernels are r2::raster_preprocess_kernel / raster_emit_hist_kernel)
    "raster.preprocess": "r2::raster_preprocess_tf_kernel",
    "raster.duplicate": "r2::(anonymous namespace)::raster_tf_scatter_kernel",
    "raster.sort": "r2::(anonymous namespace)::raster_tf_sort_kernel",
}


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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n):
    """No launcher around us but --gpus n > 1: start our own n ranks, one per GPU, and hand their exit code back."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


class StepRunner:
    """The step / exchange / drain logic of the data-parallel bench, independent of what renders: `render(k, i)` runs forward +
    backward of step k on this rank and returns the flat gradient block to exchange.  mode "sync": the all-reduce is waited for
    (stream-ordered) before the step returns; "overlap": it is waited for two steps later (double-buffered, gradients one step
    stale).  Two SYNCHRONOUS modes with two views per rank and optimiser step (no stale gradients; a step = 2 views):
    "sync2": the all-reduce of view A's block runs behind the render of view B, then B's is waited for -- two exchanges per
    step, one of them hidden; "accum2": view B's backward accumulates into view A's gradients (render(k, i, True)) and the step
    costs ONE all-reduce -- half the bytes of sync2, the same exposed time."""
    VIEWS_PER_STEP = {"sync": 1, "overlap": 1, "sync2": 2, "accum2": 2}

    def __init__(self, render, allreduce, use_comm, mode="sync"):
        self.render, self.allreduce, self.use_comm, self.mode = render, allreduce, use_comm, mode
        self.pending = [None, None]

    def step(self, k):
        if self.mode == "sync2":
            blk_a = self.render(2 * k, 0)
            h_a = self.allreduce(blk_a) if self.use_comm else None   # async: runs while view B renders
            blk_b = self.render(2 * k + 1, 1)
            if self.use_comm:
                h_a.wait()
                self.allreduce(blk_b).wait()
            return
        if self.mode == "accum2":
            self.render(2 * k, 0)
            blk = self.render(2 * k + 1, 0, True)                    # accumulates into view A's gradient block
            if self.use_comm:
                self.allreduce(blk).wait()
            return
        i = k & 1
        if self.use_comm and self.mode == "overlap" and self.pending[i] is not None:
            self.pending[i][0].wait()          # the reduction started two steps ago is done: its buffer may be reused
            self.pending[i] = None
        blk = self.render(k, i)
        if self.use_comm:
            h = self.allreduce(blk)
            if self.mode == "sync":
                h.wait()                       # stream-ordered: whatever runs next on this stream sees the reduced sum
            else:
                self.pending[i] = (h, blk)     # keep the buffer alive

    def drain(self):
        for i in range(2):
            if self.pending[i] is not None:
                self.pending[i][0].wait()
                self.pending[i] = None


def timed_regions(runner, steps, repeats, first_step, barrier, max_over_ranks):
    """`repeats` regions of exactly `steps` steps, each bracketed by barrier + synchronize; -> per-region seconds (max over ranks)."""
    out = []
    k = first_step
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        for _i in range(steps):
            runner.step(k)
            k += 1
        runner.drain()
        barrier()
        out.append(max_over_ranks(time.perf_counter() - t0))
    return out, k


def summarize(region_s, steps, world):
    med = statistics.median(region_s)
    return {"value": round(steps * world / med, 2), "ms_per_step": round(1e3 * med / steps, 4),
            "regions": len(region_s), "ms_per_step_min": round(1e3 * min(region_s) / steps, 4),
            "ms_per_step_max": round(1e3 * max(region_s) / steps, 4),
            "spread": round((max(region_s) - min(region_s)) / med, 4)}

def inv_softplus(x):
    import torch
    return torch.log(torch.expm1(x))


class TrainIteration:
    """ONE iteration of the reference's training loop (train.py:97-177) on the drop-in surface, data-parallel over the ranks of
    the process group -- what `value` (raster forward + backward only) leaves out, and where the exchange can hide:
        activations (gaussian_model.py:38-64,112-126) -> rasterize this rank's view -> L1 + 0.25 DSSIM (fused) -> backward
        -> densification statistics of the view (train.py:151-154) -> [P,11] image-gradient block: all-reduce STARTED (async)
        -> meanwhile the TV regulariser: 32^3 voxelizer query of the same patch on every rank + fused TV loss + backward;
           its gradient is identical on all ranks (replicas + same patch), so it needs no exchange
        -> wait for the all-reduce -> grad = mean of the ranks' image gradients + TV gradient -> Adam (four groups) step.
    Synchronous: every optimiser step applies this step's gradients of all ranks.  `ops` carries the renderer pieces so that
    tests/test_dist_cpu.py can run the same control flow on CPU over gloo with stand-ins."""

    def __init__(self, ops, cloud, dev, n_views, world, rank, use_comm, allreduce, fused_adam):
        import torch
        from r2_gaussian_amd import dist as r2dist
        self.ops, self.world, self.rank, self.use_comm, self.allreduce, self.n_views = ops, world, rank, use_comm, allreduce, n_views
        self.r2dist = r2dist
        sc = cloud.scales.to(dev)
        self.lo, self.hi = float(sc.min()) * 0.5, float(sc.max()) * 2.0
        y = ((sc - self.lo) / (self.hi - self.lo)).clamp(1e-6, 1 - 1e-6)
        self.leaves = [cloud.xyz.to(dev).clone().requires_grad_(True),
                       inv_softplus(cloud.density.to(dev).reshape(-1, 1).clamp_min(1e-6)).requires_grad_(True),
                       torch.log(y / (1 - y)).requires_grad_(True),
                       cloud.rotations.to(dev).clone().requires_grad_(True)]
        P = self.leaves[0].shape[0]
        lrs = (0.0002, 0.01, 0.005, 0.001)   # arguments/__init__.py:47-72
        kw = {"fused": True} if fused_adam else {}
        self.opt = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(self.leaves, lrs)], lr=0.0, eps=1e-15, **kw)
        self.m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
        self.flats = [torch.empty((P, r2dist.GRAD_WIDTH), dtype=torch.float32, device=dev) for _ in range(2)]
        self.max_radii2D = torch.zeros(P, device=dev)
        self.grad_accum = torch.zeros((P, 1), device=dev)
        self.denom = torch.zeros((P, 1), device=dev)

    def activated(self):
        import torch
        import torch.nn.functional as F
        x, d, s, r = self.leaves
        return x, F.softplus(d), torch.sigmoid(s) * (self.hi - self.lo) + self.lo, F.normalize(r)

    def step(self, k):
        ops, r2dist = self.ops, self.r2dist
        self.opt.zero_grad(set_to_none=True)
        self.m2.grad = None
        x, d, s, r = self.activated()
        vi = r2dist.view_for(k, self.n_views, rank_=self.rank, world_=self.world)
        img, radii = ops["render"](vi, x, self.m2, d, s, r)
        ops["image_loss"](img, vi).backward()
        ops["densify_stats"](radii, self.m2.grad, self.max_radii2D, self.grad_accum, self.denom)
        blk = r2dist.pack_grads(*(p.grad for p in self.leaves), out=self.flats[k & 1])
        h = self.allreduce(blk) if self.use_comm else None        # runs behind the TV branch
        for p in self.leaves:
            p.grad = None
        x, d, s, r = self.activated()
        (0.05 * ops["tv_loss"](ops["query32"](k, x, d, s, r))).backward()
        if h is not None:
            h.wait()
        for p, g in zip(self.leaves, r2dist.unpack_grads(blk)):
            p.grad.add_(g.reshape(p.shape), alpha=1.0 / self.world)
        self.opt.step()

    def drain(self):
        pass


class TrainIterationW(TrainIteration):
    """ONE optimiser step on W views rendered by THIS GPU -- the single-GPU form of the configuration DESIGN.md section 6 states
    for N = 8 ranks (W = 8 views per step, the four learning rates x W): W / V batched calls of V views each
    (GaussianRasterizerBatch: r2_raster_forward_batch / _backward_batch on the tile-first chain), the image loss per view, the
    parameter gradients of the calls accumulated by autograd, ONE TV branch, ONE Adam step.  `ops["render_batch"](b, ...)` renders
    batch b of the view set; `ops["densify_stats"]` is fed per view (the statistics are per view, train.py:151-154)."""

    def __init__(self, ops, cloud, dev, n_batches, V, W, fused_adam):
        super().__init__(ops, cloud, dev, 1, 1, 0, False, None, fused_adam)
        import torch
        self.V, self.W, self.n_batches = V, W, n_batches
        for g in self.opt.param_groups:   # the linear rule: all four group learning rates x W
            g["lr"] *= W
        P = self.leaves[0].shape[0]
        self.m2b = torch.zeros((V, P, 3), device=dev, requires_grad=True)

    def step(self, k):
        ops = self.ops
        self.opt.zero_grad(set_to_none=True)
        for c in range(self.W // self.V):
            self.m2b.grad = None
            x, d, s, r = self.activated()
            imgs, radii = ops["render_batch"]((k * (self.W // self.V) + c) % self.n_batches, x, self.m2b, d, s, r)
            loss = 0.0
            for v in range(self.V):
                loss = loss + ops["image_loss"](imgs[v:v + 1], v)
            (loss * (1.0 / self.W)).backward()   # gradients of the calls accumulate in the leaves' .grad
            for v in range(self.V):
                ops["densify_stats"](radii[v], self.m2b.grad[v], self.max_radii2D, self.grad_accum, self.denom)
        x, d, s, r = self.activated()
        (0.05 * ops["tv_loss"](ops["query32"](k, x, d, s, r))).backward()
        self.opt.step()


def stub_ops(P, HW):
    """CPU stand-ins for the renderer pieces of TrainIteration / the sharded query (R2_BENCH_STUB=1: control flow only)."""
    import torch

    def render(vi, x, m2, d, s, r):
        v = (x.sum() + m2.sum() + d.sum() + s.sum() + r.sum()) * 1e-6
        return v * torch.ones((4, 4)), torch.ones(P, dtype=torch.int32)

    class Vox:
        def __init__(self, st):
            self.st = st

        def __call__(self, means3D, opacities, scales, rotations):
            # a slab = the full volume's settings + its range of tile layers (dist.slab_settings); it is filled with the centre
            # of that range, so that the gathered volume can be checked
            t0, t1 = int(getattr(self.st, "tile_x0", 0)), int(getattr(self.st, "tile_x1", (self.st.nVoxel_x + 7) // 8))
            x0, x1 = 8 * t0, min(8 * t1, self.st.nVoxel_x)
            dv = float(self.st.sVoxel_x) / self.st.nVoxel_x
            c = float(self.st.center_x) - 0.5 * float(self.st.sVoxel_x) + (x0 + 0.5 * (x1 - x0)) * dv
            return torch.full((x1 - x0, self.st.nVoxel_y, self.st.nVoxel_z), c) + 0.0 * means3D.sum(), None
    return {"render": render, "image_loss": lambda img, vi: img.abs().mean(),
            "densify_stats": lambda radii, g2, mr, ga, dn: None,
            "query32": lambda k, x, d, s, r: (x.sum() + d.sum() + s.sum() + r.sum()) * 1e-6 * torch.ones((4, 4, 4)),
            "tv_loss": lambda vol: vol.abs().mean(), "voxelizer_cls": Vox}


def sharded_query(voxelizer_cls, settings, params, barrier, max_over_ranks, gather, reps=5, inner=4):
    """256^3 (settings) query through dist.query_sharded: every rank voxelizes its x-slab; -> median seconds per query, max over ranks."""
    from r2_gaussian_amd import dist as r2dist
    import torch
    x, d, s, r = params
    ts = []
    with torch.no_grad():
        for rep in range(reps + 1):
            barrier()
            t0 = time.perf_counter()
            for _ in range(inner):
                r2dist.query_sharded(voxelizer_cls, settings, x, d, s, r, gather=gather)
            barrier()
            dt = max_over_ranks((time.perf_counter() - t0) / inner)
            if rep:
                ts.append(dt)
    return statistics.median(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--repeats", type=int, default=0, help="timed regions of --steps steps (default: 25, or 5 when steps >= 500)")
    ap.add_argument("--gaussians", type=int, default=300000)
    ap.add_argument("--detector", type=int, default=512)
    ap.add_argument("--views", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-voxel", action="store_true")
    ap.add_argument("--no-batched", action="store_true", help="skip the batched-views section")
    ap.add_argument("--no-streams", action="store_true", help="skip the concurrent-streams section")
    ap.add_argument("--no-forward-only", action="store_true", help="skip the forward-only section")
    ap.add_argument("--no-train-iteration", action="store_true", help="skip the whole-training-iteration section")
    ap.add_argument("--no-densify-pattern", action="store_true", help="skip the render-while-P-changes section")
    ap.add_argument("--densify-steps", type=int, default=100, help="steps per Gaussian count in that section (reference: 100)")
    ap.add_argument("--cloud", default=None,
                    help="render a TRAINED, densified cloud instead of the synthetic one: a point_cloud.pickle in the reference's "
                         "layout (r2_gaussian_amd.model_io), or a recipe name of scripts/train_cloud.py (small | large: looked up / "
                         "trained through tests/trained_cloud.py).  --gaussians is then the cloud's own size")
    ap.add_argument("--headline-only", action="store_true",
                    help="the single-view step and nothing else (profiler passes: every kernel row is the headline step)")
    args = ap.parse_args()
    if args.headline_only:
        args.no_voxel = args.no_batched = args.no_streams = args.no_forward_only = args.no_cpu_baseline = True
        args.no_train_iteration = args.no_densify_pattern = True
    repeats = args.repeats or (5 if args.steps >= 500 else 25)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    from r2_gaussian_amd import dist as r2dist
    from r2_gaussian_amd import scene as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    # R2_BENCH_STUB=1 (tests/test_dist_cpu.py): exercise launcher + step/exchange/drain logic on CPU over gloo with a
    # stub in place of the renderer -- no number from such a run means anything and the line says so
    stub = os.environ.get("R2_BENCH_STUB", "0") == "1"
    if not stub and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    all_cpus = os.sched_getaffinity(0)
    pinned = None
    if stub:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        pinned = r2dist.pin_to_gpu_numa_node(local_rank)   # one process per GPU, on a slice of the GPU's own socket
    force_comm = os.environ.get("R2_BENCH_FORCE_COMM", "0") == "1"   # exercise the collective path on one GPU
    if world > 1 or force_comm:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        if stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        assert dist.get_world_size() == args.gpus, "process group size %d != --gpus %d" % (dist.get_world_size(), args.gpus)
    use_comm = dist.is_initialized()

    P, HW = args.gaussians, args.detector
    views = S.make_views(args.views, (HW, HW))
    stats = {"R": 0}
    flats = [torch.empty((P, r2dist.GRAD_WIDTH), dtype=torch.float32, device=dev) for _ in range(2)]

    if stub:
        g = torch.Generator().manual_seed(rank)

        def render(k, i, accumulate=False):
            if accumulate:
                flats[i].add_(torch.rand(P, r2dist.GRAD_WIDTH, generator=g))
            else:
                flats[i].copy_(torch.rand(P, r2dist.GRAD_WIDTH, generator=g))
            return flats[i]
        _lib = None
        cloud = S.make_cloud(P, seed=0)
    else:
        from r2_gaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib
        _lib.lib()
        cloud_info = None
        if args.cloud:
            from r2_gaussian_amd import model_io
            cpath = args.cloud
            if not os.path.isfile(cpath):
                from tests import trained_cloud as TCl      # input data only: looks the recipe up, trains it when absent
                cpath = TCl.path(args.cloud)
            with torch.no_grad():
                cx, cd, cs, cr = (t.float().contiguous() for t in model_io.activate(model_io.load_point_cloud(cpath, device="cpu")))
            cloud = S.Cloud(cx, cs, cr, cd.reshape(-1, 1))
            P = int(cx.shape[0])
            flats = [torch.empty((P, r2dist.GRAD_WIDTH), dtype=torch.float32, device=dev) for _ in range(2)]
            cloud_info = {"source": os.path.relpath(cpath, ROOT) if cpath.startswith(ROOT) else cpath, "P": P,
                          "scale_median": round(float(cs.median()), 5), "scale_min": round(float(cs.min()), 5),
                          "scale_max": round(float(cs.max()), 5)}
        else:
            cloud = S.make_cloud(P, seed=0)
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
        # the screen-space placeholder whose only role is to receive dL/dmeans2D (render_query.py:113-120): an input like
        # the parameters, resident before the timed region
        means2D = torch.zeros_like(xyz, requires_grad=True)

        def render(k, i, accumulate=False):
            vi = r2dist.view_for(k, len(views), rank_=rank, world_=world)
            img, _radii = rasterizers[vi](means3D=xyz, means2D=means2D, opacities=dens, scales=scal, rotations=rot)
            means2D.grad = None
            if not accumulate:   # accumulate: autograd adds this view's gradients into the previous view's .grad block
                for p in params:
                    p.grad = None
            img.backward(dL)
            if not use_comm:
                return None
            # the backward leaves the four parameter gradients adjacent in one buffer: reduce them where they are
            blk = r2dist.grad_block(xyz.grad, dens.grad, scal.grad, rot.grad)
            if blk is None:
                blk = r2dist.pack_grads(xyz.grad, dens.grad, scal.grad, rot.grad, out=flats[i])
            stats["zero_copy"] = blk is not flats[i]
            return blk

    def allreduce(blk):
        return r2dist.allreduce_grads(blk, average=False, async_op=True)

    def barrier():
        if world > 1:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()

    def max_over_ranks(dt):
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return dt

    if os.environ.get("R2_BENCH_NOGC", "0") == "1":
        import gc
        gc.disable()

    sync_runner = StepRunner(render, allreduce, use_comm, "sync")
    k = 0
    for _ in range(args.warmup):
        sync_runner.step(k)
        k += 1
    sync_runner.drain()

    # ---- instrumented pre-pass (untimed): which stage dominates -> that one is bracketed inside the timed regions
    DOMINANT = "raster.render_bwd"
    if not stub:
        barrier()
        _lib.profile_read(reset=True)
        _lib.profile_enable(None)
        for _ in range(min(10, max(args.steps, 1))):
            sync_runner.step(k)
            k += 1
        sync_runner.drain()
        barrier()
        pre = _lib.profile_read(reset=True)
        if pre:
            DOMINANT = max(pre.items(), key=lambda kv: kv[1][0] / max(kv[1][1], 1))[0]
        _lib.profile_enable([])
        _lib.sync_wait_stats(reset=True)

    # ---- the measurement: NO instrumentation inside these regions
    region_s, k = timed_regions(sync_runner, args.steps, repeats, k, barrier, max_over_ranks)
    dom, wait_us, wait_n = (0.0, 0), 0.0, 0
    dom_regions = None
    if not stub:
        wait_us, wait_n = _lib.sync_wait_stats(reset=True)
        # ---- the dominant kernel's launches, bracketed by HIP events on the launch stream, in regions of the same shape that
        # follow immediately (same steps, same barriers; their wall time is reported next to the un-instrumented one so that
        # the two measurements can be told apart -- round 2 bracketed the kernel inside the regions that produced `value`)
        _lib.profile_read(reset=True)
        _lib.profile_enable([DOMINANT])
        dom_s, k = timed_regions(sync_runner, args.steps, min(repeats, 5), k, barrier, max_over_ranks)
        dom = _lib.profile_read(reset=True).get(DOMINANT, (0.0, 0))
        _lib.profile_enable([])
        dom_regions = summarize(dom_s, args.steps, world)
    main_t = summarize(region_s, args.steps, world)
    overlapped = None
    if use_comm:   # the pipelined mode, labelled as such, next to the synchronous number
        ov_runner = StepRunner(render, allreduce, use_comm, "overlap")
        for _ in range(min(args.warmup, 10)):
            ov_runner.step(k)
            k += 1
        ov_runner.drain()
        ov_s, k = timed_regions(ov_runner, args.steps, repeats, k, barrier, max_over_ranks)
        overlapped = summarize(ov_s, args.steps, world)
        overlapped["note"] = "all-reduce of step k overlaps the render of step k+1: gradients are consumed one step late"
    dt_step = statistics.median(region_s) / args.steps
    two_view = None
    if use_comm:   # synchronous data parallelism with two views per rank and step (StepRunner docstring): views/s of each
        two_view = {}
        for mode2 in ("sync2", "accum2"):
            rn = StepRunner(render, allreduce, use_comm, mode2)
            for _ in range(min(args.warmup, 10)):
                rn.step(k)
                k += 1
            rn.drain()
            n2 = max(1, args.steps // 2)
            s2, k = timed_regions(rn, n2, repeats, k, barrier, max_over_ranks)
            two_view[mode2] = summarize(s2, 2 * n2, world)
        two_view["note"] = ("two views per rank and optimiser step, parameters updated once, no stale gradients; sync2 = all-reduce "
                            "of view A behind the render of view B + exposed all-reduce of B; accum2 = local accumulation + ONE "
                            "all-reduce per step")

    # ---- N > 1 (or one rank through the collective path): the synchronous modes side by side
    sync_modes = None
    if use_comm:
        sync_modes = {"sync": dict(main_t, views_per_rank_and_step=1, exchange="one all-reduce per view, fully exposed"),
                      "sync2": dict(two_view["sync2"], views_per_rank_and_step=2, exchange="two all-reduces per step, one hidden behind the second view"),
                      "accum2": dict(two_view["accum2"], views_per_rank_and_step=2, exchange="one all-reduce per step (local accumulation)"),
                      "value_is": "sync"}

    # ---- the same loop with the deferred count (opt-in mode of the library, r2_defer_count_control: the forward returns a token
    # without waiting for the device, the backward resolves it): views/s and what is left of the host's wait.  NEXT TO `value`,
    # which stays the default (waiting) mode.
    deferred = None
    if not stub and world == 1 and not use_comm and not args.headline_only:
        import ctypes as _ctq
        Lq = _lib.lib()
        Lq.r2_defer_count_control(1)
        try:
            for _ in range(10):
                sync_runner.step(k)
                k += 1
            sync_runner.drain()
            _lib.sync_wait_stats(reset=True)
            stq = (_ctq.c_longlong * 3)()
            Lq.r2_defer_count_stats(stq, 1)
            dq_s, k = timed_regions(sync_runner, args.steps, repeats, k, barrier, max_over_ranks)
            wq_us, wq_n = _lib.sync_wait_stats(reset=True)
            Lq.r2_defer_count_stats(stq, 1)
            deferred = summarize(dq_s, args.steps, 1)
            deferred.update(host_wait_us_per_step=round(wq_us / max(wq_n, 1), 1), forwards_with_token=int(stq[0]),
                            forwards_that_waited_for_a_slot=int(stq[1]), predictions_short=int(stq[2]),
                            note="R2_DEFER_COUNT mode: state sized by the prediction + 50 %, no wait in the forward, the autograd "
                                 "backward resolves the token (its wait is what host_wait_us_per_step now counts)")
        finally:
            Lq.r2_defer_count_control(0)

    # ---- one whole training iteration, data-parallel (TrainIteration): iterations/s next to the raster-only number
    train_it = None
    from r2_gaussian_amd import GaussianVoxelizationSettings
    if stub:
        ops = stub_ops(P, HW)
    elif not args.no_train_iteration:
        from r2_gaussian_amd import GaussianVoxelizer
        from r2_gaussian_amd import densify as FD
        from r2_gaussian_amd import losses as FL
        gt_img = torch.full((HW, HW), 0.05, device=dev)
        vox32_t = [GaussianVoxelizer(GaussianVoxelizationSettings(1.0, 32, 32, 32, 0.25, 0.25, 0.25, -0.3 + 0.1 * (i % 7),
                                                                  0.1 * (i % 5) - 0.2, 0.05 * (i % 9) - 0.2, False, False))
                   for i in range(16)]
        ops = {"render": lambda vi, x, m2, d, s_, r: rasterizers[vi](means3D=x, means2D=m2, opacities=d, scales=s_, rotations=r),
               "image_loss": lambda img, vi: FL.image_loss(img, gt_img, 0.25)[0],
               "densify_stats": FD.densification_stats,
               "query32": lambda k_, x, d, s_, r: vox32_t[k_ % 16](means3D=x, opacities=d, scales=s_, rotations=r)[0],
               "tv_loss": FL.tv_3d_loss, "voxelizer_cls": GaussianVoxelizer}
    else:
        ops = None
    if ops is not None and (stub or not args.no_train_iteration):
        ti = TrainIteration(ops, cloud, dev, len(views), world, rank, use_comm, allreduce, fused_adam=not stub)
        for _ in range(min(args.warmup, 10)):
            ti.step(k)
            k += 1
        nti = max(2, min(args.steps, 200))
        tr_s, k = timed_regions(ti, nti, min(repeats, 5), k, barrier, max_over_ranks)
        train_it = summarize(tr_s, nti, 1)
        train_it.update(unit="iterations/s", views_per_iteration=world, views_per_s=round(train_it["value"] * world, 2),
                        what="one optimiser step of the data-parallel trainer: activations, raster fwd+bwd of this rank's view, "
                             "fused L1+DSSIM, densification statistics, [P,11] all-reduce started right after the raster backward "
                             "and waited for after the 32^3 TV branch (voxelizer fwd+bwd + fused TV loss; same patch on every "
                             "rank: no exchange), gradient combine, torch.optim.Adam(fused) step")
        if not stub and world == 1 and not use_comm:   # ... and with the deferred count (the rasterizer forward does not wait)
            _lib.lib().r2_defer_count_control(1)
            try:
                for _ in range(6):
                    ti.step(k)
                    k += 1
                _lib.sync_wait_stats(reset=True)
                trd_s, k = timed_regions(ti, nti, min(repeats, 5), k, barrier, max_over_ranks)
                wd_us, wd_n = _lib.sync_wait_stats(reset=True)
                train_it["deferred_count"] = dict(summarize(trd_s, nti, 1), unit="iterations/s",
                                                  host_wait_us_per_iteration=round(wd_us / max(nti * min(repeats, 5), 1), 1))
            finally:
                _lib.lib().r2_defer_count_control(0)
        del ti

    # ---- the same iteration on W = 8 views per optimiser step rendered by one GPU (two batched calls of V = 4)
    train_it_w8 = None
    if ops is not None and not stub and world == 1 and not args.no_train_iteration and len(views) >= 8:
        from r2_gaussian_amd import GaussianRasterizerBatch
        BVw, Ww = 4, 8
        nbw = len(views) // BVw
        bsw = []
        for b_ in range(nbw):
            vs_ = views[b_ * BVw:(b_ + 1) * BVw]
            bsw.append(GaussianRasterizerBatch(GaussianRasterizationSettings(
                image_height=HW, image_width=HW, tanfovx=vs_[0].tanfovx, tanfovy=vs_[0].tanfovy, scale_modifier=1.0,
                viewmatrix=torch.stack([v.world_view_transform for v in vs_]).to(dev),
                projmatrix=torch.stack([v.full_proj_transform for v in vs_]).to(dev),
                campos=torch.stack([v.camera_center for v in vs_]).to(dev), prefiltered=False, mode=vs_[0].mode, debug=False)))
        ops_w = dict(ops)
        ops_w["render_batch"] = lambda b_, x, m2, d, s_, r: bsw[b_](means3D=x, means2D=m2, opacities=d, scales=s_, rotations=r)
        tiw = TrainIterationW(ops_w, cloud, dev, nbw, BVw, Ww, fused_adam=True)
        for _ in range(min(args.warmup, 6)):
            tiw.step(k)
            k += 1
        ntw = max(2, min(args.steps // 4, 50))
        trw_s, k = timed_regions(tiw, ntw, min(repeats, 5), k, barrier, max_over_ranks)
        train_it_w8 = summarize(trw_s, ntw, 1)
        train_it_w8.update(unit="iterations/s", views_per_iteration=Ww, views_per_call=BVw,
                           views_per_s=round(train_it_w8["value"] * Ww, 2),
                           what="one optimiser step on W = 8 views rendered by this GPU: 2 batched calls of 4 views (tile-first chain "
                                "over the stacked views), fused L1+DSSIM per view, gradients accumulated, densification statistics per "
                                "view, ONE 32^3 TV branch, ONE Adam(fused) step with the four learning rates x 8 -- the single-GPU form "
                                "of the configuration DESIGN.md section 6 states for N = 8 (there: one view per rank + one all-reduce)")
        del tiw

    # ---- the full-volume query sharded by x-slab over the ranks (dist.query_sharded, test.py:105-112's 256^3 query)
    sharded = None
    if use_comm and (stub or not args.no_voxel):
        nvq = 64 if stub else 256
        st_q = GaussianVoxelizationSettings(1.0, nvq, nvq, nvq, 2.0, 2.0, 2.0, 0.0, 0.0, 0.0, False, False)
        qparams = tuple(t.to(dev) for t in (cloud.xyz, cloud.density, cloud.scales, cloud.rotations))
        t_ng = sharded_query(ops["voxelizer_cls"] if ops else None, st_q, qparams, barrier, max_over_ranks, gather=False, reps=3 if stub else 5)
        t_g = sharded_query(ops["voxelizer_cls"] if ops else None, st_q, qparams, barrier, max_over_ranks, gather=True, reps=3 if stub else 5)
        sharded = {"gvoxel_per_s": round(nvq ** 3 / t_ng / 1e9, 3), "ms": round(t_ng * 1e3, 3), "volume": [nvq] * 3,
                   "slab_of_rank0": list(r2dist.slab_bounds(nvq, 0, world)),
                   "with_all_gather": {"gvoxel_per_s": round(nvq ** 3 / t_g / 1e9, 3), "ms": round(t_g * 1e3, 3)},
                   "note": "every rank voxelizes its x-slab of whole 8-voxel tile layers (independent units, no exchange); time = "
                           "max over ranks; with_all_gather assembles the full volume on every rank"}
        if stub:   # the stand-in voxelizer fills a slab with its centre: the gathered volume must be the slabs in order
            full = r2dist.query_sharded(ops["voxelizer_cls"], st_q, *qparams, gather=True)
            for r_ in range(world):
                a_, b_ = r2dist.slab_bounds(nvq, r_, world)
                want_c = -1.0 + (a_ + 0.5 * (b_ - a_)) * (2.0 / nvq)
                assert b_ <= a_ or abs(float(full[a_:b_].mean()) - want_c) < 1e-5, "gathered slabs out of order"
            sharded["gather_checked"] = True

    if stub:
        if rank == 0:
            print(json.dumps({"metric": "STUB renderer on CPU/gloo (launcher + exchange logic test only; not a measurement)",
                              "value": main_t["value"], "unit": "steps/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": main_t["ms_per_step"], "higher_is_better": True,
                              "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "stub",
                              "config": {"workload": "stub", "ranks_in_process_group": dist.get_world_size() if use_comm else 1,
                                         "comm_zero_copy": False, "parallelism": "stub"},
                              "timing": main_t, "overlapped": overlapped, "two_views_per_step": two_view,
                              "sync_modes": sync_modes, "train_iteration": train_it,
                              "voxelizer": {"sharded": sharded}}))
        if use_comm:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- forward-only views/s, same views, no autograd
    fwd_only = None
    with torch.no_grad():
        for j in range(0 if args.no_forward_only else 10):
            rasterizers[j % len(views)](means3D=xyz, means2D=means2D, opacities=dens, scales=scal, rotations=rot)
        fr = []
        nf = max(20, min(args.steps, 200))
        for _ in range(0 if args.no_forward_only else 5):
            barrier()
            t0 = time.perf_counter()
            for j in range(nf):
                vi = r2dist.view_for(j, len(views), rank_=rank, world_=world)
                rasterizers[vi](means3D=xyz, means2D=means2D, opacities=dens, scales=scal, rotations=rot)
            barrier()
            fr.append(max_over_ranks(time.perf_counter() - t0))
        fwd_only = summarize(fr, nf, world) if fr else None

    # ---- batched views (new functionality, reported NEXT TO the per-view drop-in number, never instead of it): BV views of
    # the same Gaussians per call through r2_raster_forward_batch / _backward_batch -- what a trainer that accumulates
    # several views per optimiser step calls.  Same views, same upstream gradient, forward + backward.
    batched = None
    if (rank == 0 or world > 1) and not args.no_batched:
        from r2_gaussian_amd import GaussianRasterizerBatch
        BV = 4
        nbt = len(views) // BV
        bsets = []
        for b_ in range(nbt):
            vs_ = views[b_ * BV:(b_ + 1) * BV]
            bsets.append(GaussianRasterizerBatch(GaussianRasterizationSettings(
                image_height=HW, image_width=HW, tanfovx=vs_[0].tanfovx, tanfovy=vs_[0].tanfovy, scale_modifier=1.0,
                viewmatrix=torch.stack([v.world_view_transform for v in vs_]).to(dev),
                projmatrix=torch.stack([v.full_proj_transform for v in vs_]).to(dev),
                campos=torch.stack([v.camera_center for v in vs_]).to(dev), prefiltered=False, mode=vs_[0].mode, debug=False)))
        m2b = torch.zeros((BV,) + tuple(xyz.shape), device=dev, requires_grad=True)
        dLb = dL.expand(BV, HW, HW).contiguous()

        def bstep(j):
            imgb, _rb = bsets[(j * world + rank) % nbt](means3D=xyz, means2D=m2b, opacities=dens, scales=scal, rotations=rot)
            m2b.grad = None
            for p_ in params:
                p_.grad = None
            imgb.backward(dLb)
        for j in range(2 * nbt):
            bstep(j)
        nbs = max(5, args.steps // BV)
        brs = []
        for _ in range(5 if args.steps >= 500 else 15):
            barrier()
            t0 = time.perf_counter()
            for j in range(nbs):
                bstep(j)
            barrier()
            brs.append(max_over_ranks(time.perf_counter() - t0))
        batched = summarize(brs, nbs * BV, world)
        batched.update(views_per_call=BV, note="r2_raster_forward_batch + _backward_batch, %d views per call; no gradient "
                                               "exchange in this loop; each view bit-identical to the single-view call" % BV)

    # ---- two independent views in flight: one host thread + one HIP stream each (the library's state is per host thread,
    # the compiled torch boundary releases the GIL inside the calls).  The launch- and latency-bound binning chain of one
    # view overlaps the render kernels of the other: what a trainer that accumulates the gradients of two views per
    # optimiser step can do with the UNCHANGED drop-in classes.  Reported next to `value`, never instead of it.
    concurrent = None
    from r2_gaussian_amd import _C
    if world == 1 and _C._shim() is not None and not args.no_streams:   # single-GPU runs only: an extra, kept out of the multi-rank collectives
        import threading
        NT = 2
        nsteps = max(20, min(args.steps, 400))
        # per thread: its own autograd leaves over the same storage (separate .grad), its own stream
        leaves = [[t.detach().requires_grad_(True) for t in (xyz, dens, scal, rot)] for _ in range(NT)]
        m2s = [torch.zeros_like(xyz, requires_grad=True) for _ in range(NT)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(NT)]
        gate = threading.Barrier(NT + 1)
        errs = []

        def tworker(t):
            try:
                with torch.cuda.stream(streams[t]):
                    lx, ld, ls, lr = leaves[t]

                    def one(j):
                        vi = ((j * NT + t) * world + rank) % len(views)
                        img, _r = rasterizers[vi](means3D=lx, means2D=m2s[t], opacities=ld, scales=ls, rotations=lr)
                        m2s[t].grad = None
                        for p_ in leaves[t]:
                            p_.grad = None
                        img.backward(dL)
                    for j in range(20):
                        one(j)
                    streams[t].synchronize()
                    for _rep in range(5):
                        gate.wait(timeout=180)
                        for j in range(nsteps):
                            one(j)
                        streams[t].synchronize()
                        gate.wait(timeout=180)
            except Exception as ex:   # noqa: BLE001
                errs.append(ex)
                gate.abort()

        torch.cuda.synchronize()
        ths = [threading.Thread(target=tworker, args=(t,)) for t in range(NT)]
        for th_ in ths:
            th_.start()
        crs = []
        try:
            for _rep in range(5):
                gate.wait(timeout=180)
                t0 = time.perf_counter()
                gate.wait(timeout=180)
                crs.append(max_over_ranks(time.perf_counter() - t0))
        except threading.BrokenBarrierError:
            pass
        for th_ in ths:
            th_.join()
        if not errs and len(crs) == 5:
            concurrent = summarize(crs, nsteps * NT, world)
            concurrent.update(streams=NT, note="%d host threads, one HIP stream each, independent views through the drop-in "
                                               "classes (gradients per thread, to be summed by the trainer)" % NT)
        elif errs and rank == 0:
            print("concurrent-streams section failed: %r" % (errs[0],), file=sys.stderr)

    # ---- rendering while P changes (train.py:155-168: densify / prune every 100 iterations, 50k -> 300k Gaussians over ~25
    # rounds).  Every new P is a size the tile-first chain has no history for: its prediction is seeded from the previous P's
    # instances per Gaussian.  For each of 25 growing prefixes of the cloud: K steps from cold (the first call at this P
    # included) and then K more (steady state at this P); reported: views/s of both, their ratio, and what the chain did.
    densify_pattern = None
    if world == 1 and not args.no_densify_pattern:
        import ctypes as _ctd
        Ld = _lib.lib()
        nsz, Kd = 25, args.densify_steps   # (the reference densifies every 100 iterations, train.py:155-168)
        sizes = sorted({int(round(P / 6.0 * (6.0 ** (i / (nsz - 1.0))))) for i in range(nsz)})
        subs = []
        for p_i in sizes:
            lv = [t.detach()[:p_i].clone().requires_grad_(True) for t in (xyz, dens, scal, rot)]
            subs.append((lv, torch.zeros((p_i, 3), device=dev, requires_grad=True)))

        def dstep(lv, m2_, j):
            img, _r = rasterizers[j % len(views)](means3D=lv[0], means2D=m2_, opacities=lv[1], scales=lv[2], rotations=lv[3])
            m2_.grad = None
            for p_ in lv:
                p_.grad = None
            img.backward(dL)
        st0 = (_ctd.c_longlong * 5)()
        Ld.r2_tile_first_stats(st0, 1)
        t_cold = t_warm = 0.0
        j = 0
        first_call_us = []   # per size: what the FIRST forward + backward at the new P costs (the call after a densification)
        for lv, m2_ in subs:
            for phase in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i_ in range(Kd):
                    dstep(lv, m2_, j)
                    j += 1
                    if phase == 0 and i_ == 0:
                        torch.cuda.synchronize()
                        first_call_us.append((time.perf_counter() - t0) * 1e6)
                torch.cuda.synchronize()
                if phase == 0:
                    t_cold += time.perf_counter() - t0
                else:
                    t_warm += time.perf_counter() - t0
        st1 = (_ctd.c_longlong * 5)()
        Ld.r2_tile_first_stats(st1, 0)
        nd = len(subs) * Kd
        densify_pattern = {"sizes": [sizes[0], sizes[-1], len(sizes)], "steps_per_size": Kd,
                           "views_per_s_including_first_calls": round(nd / t_cold, 1), "views_per_s_steady": round(nd / t_warm, 1),
                           "ratio": round(t_warm / t_cold, 4),
                           "first_call_us": {"median": round(statistics.median(first_call_us), 1), "max": round(max(first_call_us), 1),
                                             "per_size": [round(u, 1) for u in first_call_us]},
                           "tile_first": {"taken": int(st1[0]), "general_chain": int(st1[1]), "second_pass": int(st1[2]),
                                          "thin_rerender": int(st1[3]), "seeded_from_previous_P": int(st1[4])},
                           "note": "forward + backward through the drop-in classes on %d growing prefixes of the cloud, %d steps from "
                                   "cold (first call at the new P included) then %d steady ones per size" % (len(sizes), Kd, Kd)}
        del subs

    # ---- instrumented pass: per-stage breakdown (not part of `value`)
    _lib.profile_enable(None)
    for _ in range(min(args.steps, 50)):
        sync_runner.step(k)
        k += 1
    sync_runner.drain()
    torch.cuda.synchronize()
    prof = _lib.profile_read(reset=True)
    _lib.profile_enable([])
    # num_rendered of the measured views via the C mirror
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
    # what the cloud looks like to the binning chain: tile-list lengths, and how often the hinted depth order overflowed its
    # buckets (whole-call fallback) / the thin-Gaussian render variant was compiled in, over the measured views
    import ctypes as _ct
    import numpy as _np
    Lc = _lib.lib()
    lens, n_over, n_thin, n_hinted, n_tf = [], 0, 0, 0, 0
    with torch.no_grad():
        for vi in range(0, len(views), max(1, len(views) // 10)):
            s = settings[vi]
            r_ = _C.rasterize_gaussians(xyz, dens, scal, rot, 1.0, e, s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, HW, HW,
                                        s.campos, False, s.mode, False)
            torch.cuda.synchronize()
            bid = _ct.c_int(-1)
            off = Lc.r2_raster_state_offset(15, P, r_[0], HW, HW, _ct.byref(bid))
            hwd = r_[3 + bid.value][off:off + 32].cpu().numpy().view(_np.uint32)
            n_over += int(hwd[1] != 0)
            n_thin += int(hwd[2] != 0)
            n_tf += int(hwd[3] == 0x71FE)                      # the tile-first binning chain ran (csrc/raster_tilefirst.hip)
            n_hinted += int(hwd[7] != 0 and hwd[3] != 0x71FE)  # the general chain with a hinted depth order
            off = Lc.r2_raster_state_offset(6, P, r_[0], HW, HW, _ct.byref(bid))
            rg_ = r_[3 + bid.value][off:off + 8 * T].cpu().numpy().view(_np.uint32).reshape(T, 2).astype(_np.int64)
            lens.append(rg_[:, 1] - rg_[:, 0])
    lens = _np.concatenate(lens)
    cloud_stats = {"views_sampled": len(Rl), "instances_per_gaussian": round(R / max(P, 1), 2),
                   "tile_list_len": {"p50": float(_np.percentile(lens, 50)), "p90": float(_np.percentile(lens, 90)),
                                     "p99": float(_np.percentile(lens, 99)), "max": int(lens.max())},
                   "tile_first_chain": n_tf, "depth_hint_used": n_hinted, "depth_hint_overflow": n_over, "thin_variant_ANY4": n_thin}
    if cloud_info:
        cloud_stats.update(cloud_info)
    kernels = {}
    for name, (ms, cnt) in sorted(prof.items()):
        us = 1e3 * ms / cnt
        b = algorithmic_bytes(name, P, R, T, N)
        gbps = b / (us * 1e-6) / 1e9 if us > 0 else 0.0
        kernels[name] = {"us": round(us, 2), "alg_MB": round(b / 1e6, 2), "GBps": round(gbps, 1),
                         "frac": round(gbps / HBM_PEAK_GBS, 4)}
    dom_us = 1e3 * dom[0] / max(dom[1], 1)
    dom_bytes = algorithmic_bytes(DOMINANT, P, R, T, N)
    achieved = dom_bytes / (dom_us * 1e-6) / 1e9 if dom_us > 0 else 0.0
    total_bytes = 492 * P + 108 * R + 24 * T + 16 * N

    # ---- voxelizer GVoxel/s (second half of the BASELINE metric): 300k Gaussians on the 256^3 volume
    gvox = None
    if not args.no_voxel and rank == 0:
        with torch.no_grad():
            va = (xyz, dens, scal, rot, 1.0, e, 256, 256, 256, 2.0, 2.0, 2.0, 0.0, 0.0, 0.0, False, False)
            import ctypes as _ct
            vstat = (_ct.c_longlong * 3)()
            _lib.lib().r2_voxel_sticks_stats(vstat, 1)
            paths_before_vox = _lib.path_stats()
            for _ in range(3):
                R3 = _C.voxelize_gaussians(*va)[0]
            tvs = []
            nv = 10
            for _ in range(5):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _j in range(nv):
                    _C.voxelize_gaussians(*va)
                torch.cuda.synchronize()
                tvs.append((time.perf_counter() - t1) / nv)
            tv = statistics.median(tvs)
            _lib.profile_enable(None)       # per-stage breakdown of the same call (not part of the timing above)
            for _ in range(5):
                _C.voxelize_gaussians(*va)
            torch.cuda.synchronize()
            vprof = _lib.profile_read(reset=True)
            _lib.profile_enable([])
            _lib.lib().r2_voxel_sticks_stats(vstat, 0)
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
        for i in range(30):
            tv_step(i)
        ttvs = []
        for _ in range(5):   # median of 5 x 100 iterations (one cold sample moved this number by 70 % in round 3)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for i in range(100):
                tv_step(i)
            torch.cuda.synchronize()
            ttvs.append((time.perf_counter() - t2) / 100)
        ttv = statistics.median(ttvs)
        vbytes = 168 * P + 88 * R3 + 16 * 32768 + 8 * 256 ** 3
        gvox = {"gvoxel_per_s": round(256 ** 3 / tv / 1e9, 3), "ms": round(tv * 1e3, 3), "ms_min": round(min(tvs) * 1e3, 3),
                "R3": int(R3), "alg_MB": round(vbytes / 1e6, 1), "hbm_frac": round(vbytes / tv / 1e9 / HBM_PEAK_GBS, 4),
                "stages_us": {k_: round(1e3 * ms / cnt, 1) for k_, (ms, cnt) in sorted(vprof.items()) if k_.startswith("voxel.")},
                # which binning chain the 256^3 queries above took: csrc/voxel_sticks.hip (stick-first: no global sort) or the general
                # one (depth order + radix passes); on the stick chain the stages are preprocess = cull + count, scan = column scan +
                # render records (one launch), duplicate = scatter, sort = per-list sort, ranges = work list
                "binning": {"stick_chain_calls": int(vstat[0]), "left_for_general_chain": int(vstat[1]), "general_chain_calls": int(vstat[2]),
                            # the same from the library's dispatch table (csrc/dispatch.hpp): chain taken / reason of every hand-over
                            "paths": {k_: v_ - paths_before_vox.get(k_, 0) for k_, v_ in _lib.path_stats().items()
                                      if k_.startswith("voxel.") and v_ != paths_before_vox.get(k_, 0)}},
                "tv_patch_32cube_fwd_bwd_us": round(ttv * 1e6, 1), "tv_patch_us_min": round(min(ttvs) * 1e6, 1),
                "tv_patch_us_max": round(max(ttvs) * 1e6, 1)}

    # ---- simple-knn (distCUDA2, gaussian_model.py:145-150: called once per run, on the initial points): timed
    # at the initial-cloud sizes of configs B / headline / E
    knn_t = None
    if rank == 0 and not args.no_voxel:
        from r2_gaussian_amd import distCUDA2
        knn_t = {}
        for n_ in (50000, 300000, 1000000):
            pts = S.make_cloud(n_, seed=1).xyz.to(dev)
            distCUDA2(pts)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            distCUDA2(pts)
            torch.cuda.synchronize()
            knn_t[str(n_)] = round((time.perf_counter() - t3) * 1e3, 3)
        knn_t["unit"] = "ms per call incl. its workspace allocation (exact uniform-grid 3-NN search, csrc/knn.hip; the exhaustive " \
                        "O(P^2) kernel it replaced above 4096 points: 2.0 / 33 / 313 ms)"

    # ---- CPU baseline + parity self-check: the oracle on the host cores, ONE view; the same view's GPU result is checked
    # against it before the line is printed
    cpu, parity = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        os.sched_setaffinity(0, all_cpus)   # the CPU baseline may use every host core again
        from oracle import oracle as O
        from oracle import parity as Pz
        O.lib()
        v = views[0]
        xn, dn, sn, qn = (t.detach().cpu().numpy() for t in (xyz, dens, scal, rot))
        vm, pm = v.world_view_transform.numpy(), v.full_proj_transform.numpy()
        dLn = dL.cpu().numpy()
        tc = time.perf_counter()
        st = O.raster_forward(xn, dn, sn, qn, 1.0, None, vm, pm, v.tanfovx, v.tanfovy, HW, HW, v.mode)
        O.raster_backward(st, xn, sn, qn, 1.0, None, vm, pm, v.tanfovx, v.tanfovy, dLn, acc64=False)
        tc = time.perf_counter() - tc
        nthr = int(O.lib().r2o_num_threads())
        cpu = {"value": round(1.0 / tc, 4), "unit": "views/s", "cores": nthr, "kind": "port",
               "what": "C/OpenMP port of the reference algorithm (oracle/r2_oracle.c), %d threads: a stronger CPU baseline than "
                       "the pure-PyTorch evaluation north_star mentions, which is reported beside it (pure_pytorch_config_a)" % nthr,
               "sample": "1 view fwd+bwd of the same workload (%dk Gaussians, %d^2, R=%d)" % (P // 1000, HW, st["num_rendered"])}
        # the pure-PyTorch CPU evaluation north_star names: BASELINE configs[0] (5k Gaussians, 64^2, 10 views + a 64^3 query),
        # forward + autograd backward, run fully (oracle/torch_baseline.py, checked against the oracle in tests/).  In a FRESH
        # process with a bounded wait: this one's OpenMP workers were created while it was pinned to a slice of the GPU's socket
        # and keep that affinity -- torch CPU kernels with one thread per host core would crawl on them (minutes, not seconds)
        try:
            nthr = max(1, min(32, len(all_cpus)))
            code = ("import sys, json, torch; sys.path.insert(0, %r); torch.set_num_threads(%d); "
                    "from oracle import torch_baseline as TB; TB.config_a(n_gaussians=500, detector=32, n_views=1, n_voxel=16); "
                    "t, n, _i, _v = TB.config_a(); print(json.dumps({'seconds': t, 'views': n, 'threads': torch.get_num_threads()}))"
                    % (ROOT, nthr))
            env = dict(os.environ, OMP_NUM_THREADS=str(nthr), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=150, env=env, cwd=ROOT)
            ja = json.loads(r.stdout.strip().splitlines()[-1])
            cpu["pure_pytorch_config_a"] = {
                "seconds": round(ja["seconds"], 3), "views_per_s": round(ja["views"] / ja["seconds"], 3), "threads": ja["threads"],
                "what": "oracle/torch_baseline.py in a fresh process: 5k Gaussians, 64^2 detector, 10 views forward + autograd "
                        "backward, plus one 64^3 volume query forward + backward; float32, tile-exact lists"}
        except Exception as ex:   # a reported extra: never fails (or stalls) the bench line
            cpu["pure_pytorch_config_a"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}
        # ... and on THIS workload ("for B / C / E time one view and extrapolate"): one view forward + autograd
        # backward through the same pure-PyTorch evaluation, same fresh-process / bounded-wait arrangement
        if not args.cloud:
            try:
                code = ("import sys, json, torch; sys.path.insert(0, %r); torch.set_num_threads(%d); "
                        "from oracle import torch_baseline as TB; t, nt = TB.one_view(%d, %d, %d); "
                        "print(json.dumps({'seconds': t, 'tiles': nt, 'threads': torch.get_num_threads()}))"
                        % (ROOT, nthr, P, HW, args.views))
                r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
                jh = json.loads(r.stdout.strip().splitlines()[-1])
                cpu["pure_pytorch_this_workload"] = {
                    "seconds_per_view": round(jh["seconds"], 2), "views_per_s": round(1.0 / jh["seconds"], 4),
                    "threads": jh["threads"],
                    "what": "oracle/torch_baseline.py, ONE view (view 0) forward + autograd backward of this workload "
                            "(%dk Gaussians, %d^2), timed once in a fresh process; views/s is the extrapolation 1 / seconds"
                            % (round(P / 1000), HW)}
            except Exception as ex:
                cpu["pure_pytorch_this_workload"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}
        s0 = settings[0]
        with torch.no_grad():
            Rg, color, radii, gb, bb, ib = _C.rasterize_gaussians(xyz, dens, scal, rot, 1.0, e, s0.viewmatrix, s0.projmatrix,
                                                                  s0.tanfovx, s0.tanfovy, HW, HW, s0.campos, False, s0.mode, False)
            res = _C.rasterize_gaussians_backward(xyz, radii, scal, rot, 1.0, e, s0.viewmatrix, s0.projmatrix, s0.tanfovx,
                                                  s0.tanfovy, dL, s0.campos, gb, Rg, bb, ib, s0.mode, False)
        torch.cuda.synchronize()
        names = ["dL_dmeans2D", "dL_dopacity", "dL_dmu", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations"]
        gh = {n_: t.cpu().numpy() for n_, t in zip(names, res)}
        parity = {"view": 0, "rtol": Pz.RTOL, "ok": False}
        try:
            assert Rg == st["num_rendered"], "num_rendered %d != oracle %d" % (Rg, st["num_rendered"])
            assert (radii.cpu().numpy() == st["radii"]).all(), "radii differ from the oracle"
            budget, _nb = O.raster_forward_audit(st)
            si = Pz.image_parity(color.cpu().numpy(), st["color"], budget)
            sg = Pz.raster_grad_parity(O, st, dLn, gh, xn, sn, qn, 1.0, None, vm, pm, v.tanfovx, v.tanfovy)
            parity.update({"ok": True, "num_rendered_equal": True, "radii_equal": True,
                           "max_rel_err": si["max_rel_err"], "n_flip_pixels": si["n_flips"],
                           "n_flip_candidate_pixels": si["n_flip_candidates"],
                           "grad_max_err_over_tol": round(max(sg[n_]["max_err_over_tol"] for n_ in names if n_ in sg), 5),
                           "grad_max_err_over_scale": max(sg[n_]["max_err_over_scale_unflagged"] for n_ in names if n_ in sg),
                           "n_flip_candidate_gaussians": sg["n_flip_candidates"]})
        except AssertionError as ex:
            parity["error"] = str(ex)[:1500]

    # HBM traffic of the dominant kernel from the TCC counters (FETCH_SIZE / WRITE_SIZE, one counter per rocprofv3 pass:
    # scripts/gpu_pmc.sh; the summary it writes is committed under profiles/).  bench.py cannot collect PMCs itself; the
    # summary is accepted only when it was collected on the kernel sources this library was built from.
    traffic, traffic_note, valu = None, None, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path):
        try:
            from r2_gaussian_amd import build as r2build
            pmc = json.load(open(pmc_path))
            have, want = pmc.get("_meta", {}).get("source_sha"), r2build.source_hash()
            kq = pmc.get(STAGE_KERNEL.get(DOMINANT, ""), {})
            if (P, HW, args.views) != (300000, 512, 50) or args.cloud:
                traffic_note = "profiles/pmc_latest.json holds the counters of the headline workload (300k Gaussians, 512^2): " \
                               "not applicable to this one"
            elif have != want:
                traffic_note = "profiles/pmc_latest.json was collected on other kernel sources (%s != %s): not used" % (
                    str(have)[:12], want[:12])
            elif "FETCH_SIZE" in kq and "WRITE_SIZE" in kq:
                traffic = int((kq["FETCH_SIZE"] + kq["WRITE_SIZE"]) * 1024)   # counters are in KB, per launch
                if "SQ_INSTS_VALU" in kq:
                    # what actually bounds this kernel: its VALU instruction stream against the rate the chip
                    # retires independent f32 instructions at with four waves per SIMD (scripts/ubench_valu.hip,
                    # profiles/r06_ubench_valu.txt: 2.8 cycles per wave-instruction per SIMD at 2.4 GHz)
                    floor_us = kq["SQ_INSTS_VALU"] / VALU_SIMDS * VALU_CYCLES_PER_INST / VALU_CLOCK_GHZ / 1e3
                    valu = {"insts_per_launch": int(kq["SQ_INSTS_VALU"]), "cycles_per_inst_floor": VALU_CYCLES_PER_INST,
                            "clock_ghz": VALU_CLOCK_GHZ, "simds": VALU_SIMDS, "floor_us": round(floor_us, 2),
                            "note": "SQ_INSTS_VALU per launch (profiles/pmc_latest.json) x measured issue floor; frac = floor_us / "
                                    "us_per_launch is filled in below"}
                traffic_note = "FETCH_SIZE + WRITE_SIZE per launch, separate rocprofv3 --pmc passes (profiles/pmc_latest.json, " \
                               "sources %s); gfx950 FETCH_SIZE under-reports 16 B/lane streaming reads by 2x (uncorrected " \
                               "here: the kernel gathers)" % want[:12]
        except Exception as ex:
            traffic_note = "pmc summary unreadable: %r" % (ex,)

    # cpu_baseline, top level = the evaluation north_star names (pure PyTorch on the host cores, ONE view of this workload timed and
    # extrapolated); the C/OpenMP port -- the stronger baseline, and the oracle the parity check above used -- beside it
    cpu_note = None
    if cpu is not None:
        pt = cpu.get("pure_pytorch_this_workload")
        if isinstance(pt, dict) and "views_per_s" in pt:
            port = {k_: cpu[k_] for k_ in ("value", "unit", "cores", "kind", "what", "sample")}
            cpu = {"value": pt["views_per_s"], "unit": "views/s", "cores": pt["threads"], "kind": "port",
                   "implementation": "pure-PyTorch float32 evaluation of the same math (oracle/torch_baseline.py, tile-exact lists, "
                                     "forward + autograd backward), fresh process on the host cores",
                   "sample": "1 view (view 0) fwd+bwd of the same workload, %.2f s; views/s = 1 / seconds" % pt["seconds_per_view"],
                   "c_openmp_port": port, "pure_pytorch_config_a": cpu.get("pure_pytorch_config_a")}
        else:
            cpu["note"] = "top level = the C/OpenMP port: the pure-PyTorch evaluation of this workload was not available (%s)" % (
                "trained cloud" if args.cloud else str(pt)[:120])
    elif world > 1:
        cpu_note = "omitted for N > 1 (the CPU baseline is timed on rank 0 at N = 1 only, as the bench contract says)"
    if rank == 0:
        if world > 1 or use_comm:
            par = ("view-sharded dp%d: one view per rank and optimiser step, ONE RCCL all-reduce of the [P,11] gradient block per step "
                   "(%s), %d ranks in the process group; value = mode 'sync' (the all-reduce of step k is waited for before step "
                   "k+1: fully exposed); sync2 / accum2 (two views per rank and step, exchange hidden / halved) in sync_modes" % (
                       world, "in place, zero-copy" if stats.get("zero_copy") else "packed copy", dist.get_world_size()))
        else:
            par = "single GPU"
        out = {
            "metric": "rasterized X-ray views/sec (fwd+bwd) at %dk Gaussians, %d^2 cone-beam detector" % (round(P / 1000), HW),
            "value": main_t["value"], "unit": "views/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": main_t["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("synthetic 0_chest_cone-like cone-beam set: %d Gaussians (seed 0), %dx%d detector, "
                                    "%d views, DSD 7 / DSO 5" % (P, HW, HW, args.views)) if not args.cloud else
                                   ("TRAINED densified cloud (%s): %d Gaussians, %dx%d detector, %d views of the synthetic "
                                    "cone-beam set it was trained on, DSD 7 / DSO 5" % (cloud_info["source"], P, HW, HW, args.views)),
                       "num_rendered": R, "parallelism": par,
                       "ranks_in_process_group": dist.get_world_size() if use_comm else 1,
                       "comm_zero_copy": stats.get("zero_copy") if use_comm else None},
            "cloud_stats": cloud_stats,
            "paths": {k_: v_ for k_, v_ in _lib.path_stats().items() if v_},   # every forward of this process, by chain and reason
            "timing": dict(main_t, note="median of %d regions of exactly %d steps, each between barrier + synchronize, "
                                        "max over ranks" % (repeats, args.steps)),
            "overlapped": overlapped,
            "two_views_per_step": two_view,
            "sync_modes": sync_modes,
            "deferred_count": deferred,
            "train_iteration": train_it,
            "train_iteration_w8": train_it_w8,
            "densify_pattern": densify_pattern,
            "forward_only": fwd_only,
            "batched": batched,
            "concurrent_streams": concurrent,
            "roofline": {"bound": "hbm", "kernel": DOMINANT, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_note": traffic_note, "us_per_launch": round(dom_us, 2), "launches_timed": int(dom[1]),
                         "timed_in": "separate regions of the same shape right after the ones that produce `value` "
                                     "(HIP events around this kernel's launches only)",
                         "instrumented_regions": dom_regions,
                         "alg_bytes_per_launch": dom_bytes,
                         "pipeline_frac": round(total_bytes / dt_step / 1e9 / HBM_PEAK_GBS, 4),
                         "pipeline_alg_bytes": total_bytes,
                         "valu": (dict(valu, frac=round(valu["floor_us"] / dom_us, 3)) if valu and dom_us > 0 else None)},
            "cpu_baseline": cpu,
            "cpu_baseline_note": cpu_note,
            "parity_checked": parity,
            "comm_zero_copy": stats.get("zero_copy") if use_comm else None,
            # host time per step spent waiting for num_rendered at the forward's sync: large = GPU-bound step
            "host_wait_us_per_step": round(wait_us / max(wait_n, 1), 1),
            "host_cpus_pinned": len(pinned) if pinned else None,
            "kernels": kernels,
            "voxelizer": (dict(gvox or {}, sharded=sharded) if (gvox or sharded) else None),
            "simple_knn_ms": knn_t,
        }
        print(json.dumps(out))
    if dist.is_initialized():
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()
    if parity is not None and not parity["ok"]:
        sys.stderr.write("bench.py: PARITY CHECK FAILED: %s\n" % parity.get("error"))
        raise SystemExit(3)


if __name__ == "__main__":
    main()
