"""A miniature R2-Gaussian trainer for the 3D-PSNR parity study (TEST INFRASTRUCTURE: not part of the product; the product is
the kernels behind the drop-in packages, on which the reference's own train.py / GaussianModel run unmodified).

It restates, in compact form, exactly the parts of the reference that decide the optimisation trajectory:
  * the training iteration                       train.py:97-177
  * parameters, activations, Adam groups, LR     r2_gaussian/gaussian/gaussian_model.py:38-64,112-126,133-164,188-254
  * densify (clone / split) and prune            r2_gaussian/gaussian/gaussian_model.py:320-556
  * L1 + 0.25 (1 - SSIM) + 0.05 TV(32^3)         r2_gaussian/utils/loss_utils.py:19-104, arguments/__init__.py:47-72
  * render() / query() glue                      r2_gaussian/gaussian/render_query.py:27-160
  * initialisation from a volume                 initialize_pcd.py:64-89, gaussian_model.py:133-164
  * 3D PSNR                                      r2_gaussian/utils/image_utils.py:90-104
and runs them on either backend:
  "hip"    -- the drop-in packages (xray_gaussian_rasterization_voxelization, simple_knn) on cuda:0: the MI355X kernels;
  "oracle" -- test-only torch.autograd.Functions around the CPU oracle (oracle/r2_oracle.c): the reference's arithmetic.
All randomness (view order, TV patch centres, split samples) comes from seeded CPU generators, so both backends see the same
stream as long as they take the same densification decisions.

The synthetic case ("0_chest_cone-synthetic"): a hidden set of GT Gaussians (seed 2) rendered by the ORACLE
gives the training projections, its oracle voxelization is vol_gt, the initial points are sampled from vol_gt > 0.05 with
density x 0.15.
"""
import math
import random
import time

import numpy as np
import torch
import torch.nn.functional as F

from r2_gaussian_amd import scene as S


# ---------------------------------------------------------------------------------------------- configuration
class Opt:
    """OptimizationParams of the reference (arguments/__init__.py:44-72), iteration counts scaled by the caller."""
    iterations = 30000
    position_lr_init, position_lr_final = 0.0002, 0.00002
    density_lr_init, density_lr_final = 0.01, 0.001
    scaling_lr_init, scaling_lr_final = 0.005, 0.0005
    rotation_lr_init, rotation_lr_final = 0.001, 0.0001
    lambda_dssim, lambda_tv, tv_vol_size = 0.25, 0.05, 32
    density_min_threshold = 0.00001
    densification_interval, densify_from_iter, densify_until_iter = 100, 500, 15000
    densify_grad_threshold, densify_scale_threshold = 5.0e-5, 0.1
    max_num_gaussians = 500000
    max_screen_size, max_scale = None, None      # optional prune thresholds (arguments/__init__.py:67-68); max_scale in % of volume
    scale_min, scale_max = 0.0005, 0.5

    def __init__(self, **kw):
        for k, v in kw.items():
            if not hasattr(Opt, k):
                raise AttributeError(k)
            setattr(self, k, v)
        self.lr_max_steps = kw.get("iterations", self.iterations)   # *_lr_max_steps follow the run length


def expon_lr(lr_init, lr_final, max_steps):
    """log-linear interpolation lr_init -> lr_final (utils/gaussian_utils.py:13-46, no delay)."""
    def f(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        t = min(max(step / max_steps, 0.0), 1.0)
        return math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)
    return f


# ---------------------------------------------------------------------------------------------- oracle backend
class _OracleRaster(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, opacities, scales, rotations, v):
        from oracle import oracle as O
        a = [t.detach().cpu().numpy() for t in (means3D, opacities, scales, rotations)]
        vm, pm = v.world_view_transform.numpy(), v.full_proj_transform.numpy()
        st = O.raster_forward(a[0], a[1], a[2], a[3], 1.0, None, vm, pm, v.tanfovx, v.tanfovy, v.image_height, v.image_width,
                              v.mode)
        ctx.st, ctx.a, ctx.v = st, a, v
        radii = torch.from_numpy(st["radii"].copy())
        ctx.mark_non_differentiable(radii)
        return torch.from_numpy(st["color"].copy()), radii

    @staticmethod
    def backward(ctx, g, _):
        from oracle import oracle as O
        v, a = ctx.v, ctx.a
        vm, pm = v.world_view_transform.numpy(), v.full_proj_transform.numpy()
        r = O.raster_backward(ctx.st, a[0], a[2], a[3], 1.0, None, vm, pm, v.tanfovx, v.tanfovy, g.contiguous().numpy(),
                              acc64=False)   # float accumulation, like the reference's atomics
        t = torch.from_numpy
        return t(r["dL_dmeans3D"]), t(r["dL_dmeans2D"]), t(r["dL_dopacity"]), t(r["dL_dscales"]), t(r["dL_drotations"]), None


class _OracleVoxel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, opacities, scales, rotations, geo):
        from oracle import oracle as O
        a = [t.detach().cpu().numpy() for t in (means3D, opacities, scales, rotations)]
        n, s, c = geo
        st = O.voxel_forward(a[0], a[1], a[2], a[3], 1.0, None, n, s, c)
        ctx.st, ctx.a = st, a
        return torch.from_numpy(st["vol"].copy())

    @staticmethod
    def backward(ctx, g):
        from oracle import oracle as O
        a = ctx.a
        r = O.voxel_backward(ctx.st, a[2], a[3], 1.0, None, g.contiguous().numpy(), acc64=False)
        t = torch.from_numpy
        return t(r["dL_dmeans3D"]), t(r["dL_dopacity"]), t(r["dL_dscales"]), t(r["dL_drotations"]), None


class Backend:
    """render() / query() / distCUDA2 of one backend (render_query.py:27-160 semantics)."""

    def __init__(self, name):
        assert name in ("hip", "oracle")
        self.name = name
        self.device = torch.device("cuda:0") if name == "hip" else torch.device("cpu")
        self._settings = {}

    def render(self, v, xyz, dens, scales, rot):
        screen = torch.zeros_like(xyz, requires_grad=True) + 0
        screen.retain_grad()
        if self.name == "oracle":
            img, radii = _OracleRaster.apply(xyz, screen, dens, scales, rot, v)
        else:
            from xray_gaussian_rasterization_voxelization import GaussianRasterizationSettings, GaussianRasterizer
            rs = self._settings.get(id(v))
            if rs is None:
                d = self.device
                rs = self._settings[id(v)] = GaussianRasterizationSettings(
                    image_height=v.image_height, image_width=v.image_width, tanfovx=v.tanfovx, tanfovy=v.tanfovy,
                    scale_modifier=1.0, viewmatrix=v.world_view_transform.to(d), projmatrix=v.full_proj_transform.to(d),
                    campos=v.camera_center.to(d), prefiltered=False, mode=v.mode, debug=False)
            img, radii = GaussianRasterizer(raster_settings=rs)(means3D=xyz, means2D=screen, opacities=dens, scales=scales,
                                                                rotations=rot, cov3D_precomp=None)
        return dict(render=img, viewspace_points=screen, visibility_filter=radii > 0, radii=radii)

    def query(self, xyz, dens, scales, rot, center, nVoxel, sVoxel):
        n = tuple(int(x) for x in nVoxel)
        s = tuple(float(x) for x in sVoxel)
        c = tuple(float(x) for x in center)
        if self.name == "oracle":
            return _OracleVoxel.apply(xyz, dens, scales, rot, (n, s, c))
        from xray_gaussian_rasterization_voxelization import GaussianVoxelizationSettings, GaussianVoxelizer
        vs = GaussianVoxelizationSettings(scale_modifier=1.0, nVoxel_x=n[0], nVoxel_y=n[1], nVoxel_z=n[2], sVoxel_x=s[0],
                                          sVoxel_y=s[1], sVoxel_z=s[2], center_x=c[0], center_y=c[1], center_z=c[2],
                                          prefiltered=False, debug=False)
        vol, _radii = GaussianVoxelizer(voxel_settings=vs)(means3D=xyz, opacities=dens, scales=scales, rotations=rot,
                                                           cov3D_precomp=None)
        return vol

    def dist2(self, pts):
        if self.name == "oracle":
            from oracle import oracle as O
            return torch.from_numpy(O.knn_dist2(pts.cpu().numpy()))
        from simple_knn._C import distCUDA2
        return distCUDA2(pts)


# ---------------------------------------------------------------------------------------------- the synthetic case
class Case:
    """GT Gaussians -> training projections (oracle render) + vol_gt (oracle voxelization) + initial points."""

    def __init__(self, detector=128, n_vol=64, n_views=50, p_gt=20000, n_init=5000, seed=2):
        from oracle import oracle as O
        self.scanner = dict(S.CONE_BEAM, nVoxel=[n_vol] * 3)
        self.views = S.make_views(n_views, (detector, detector))
        gt = S.make_cloud(p_gt, seed=seed)
        a = (gt.xyz.numpy(), gt.density.numpy(), gt.scales.numpy(), gt.rotations.numpy())
        self.projs = []
        for v in self.views:
            st = O.raster_forward(a[0], a[1], a[2], a[3], 1.0, None, v.world_view_transform.numpy(),
                                  v.full_proj_transform.numpy(), v.tanfovx, v.tanfovy, detector, detector, v.mode)
            self.projs.append(torch.from_numpy(st["color"].copy()))
        self.nVoxel, self.sVoxel, self.center = (n_vol,) * 3, (2.0, 2.0, 2.0), (0.0, 0.0, 0.0)
        self.vol_gt = torch.from_numpy(O.voxel_forward(a[0], a[1], a[2], a[3], 1.0, None, self.nVoxel, self.sVoxel,
                                                       self.center)["vol"].copy())
        # initial points: n_init voxels with vol_gt > 0.05, density x 0.15 (initialize_pcd.py:67-86)
        rng = np.random.default_rng(seed + 1)
        idx = np.argwhere(self.vol_gt.numpy() > 0.05)
        pick = idx[rng.choice(len(idx), n_init, replace=False)]
        dV = np.array(self.sVoxel) / np.array(self.nVoxel)
        self.init_xyz = torch.tensor(pick * dV - np.array(self.sVoxel) / 2 + np.array(self.center), dtype=torch.float32)
        self.init_density = torch.tensor(self.vol_gt.numpy()[pick[:, 0], pick[:, 1], pick[:, 2]] * 0.15, dtype=torch.float32)
        self.bbox = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
        self.dVoxel = torch.tensor(dV, dtype=torch.float32)


# ---------------------------------------------------------------------------------------------- model
class Model:
    """GaussianModel in miniature (gaussian_model.py): raw parameters, activations, Adam with four groups, densify/prune."""
    NAMES = ("xyz", "density", "scaling", "rotation")

    def __init__(self, case, opt, backend, gen):
        self.opt, self.be, self.gen = opt, backend, gen
        dev = backend.device
        self.lo, self.hi = opt.scale_min * 2.0, opt.scale_max * 2.0       # scale bound * volume_to_world (train.py:59-61)
        xyz = case.init_xyz.to(dev)
        dist = torch.sqrt(torch.clamp_min(backend.dist2(xyz).to(dev), 0.001 ** 2))
        dist = torch.clamp(dist, self.lo + 1e-7, self.hi - 1e-7)
        self.p = {
            "xyz": xyz.clone().requires_grad_(True),
            "density": self.inv_softplus(case.init_density.to(dev))[:, None].contiguous().requires_grad_(True),
            "scaling": self.scaling_inv(dist)[:, None].repeat(1, 3).contiguous().requires_grad_(True),
            "rotation": torch.tensor([1.0, 0, 0, 0], device=dev).repeat(xyz.shape[0], 1).requires_grad_(True),
        }
        self.max_radii2D = torch.zeros(xyz.shape[0], device=dev)
        self.lr = {"xyz": expon_lr(opt.position_lr_init, opt.position_lr_final, opt.lr_max_steps),
                   "density": expon_lr(opt.density_lr_init, opt.density_lr_final, opt.lr_max_steps),
                   "scaling": expon_lr(opt.scaling_lr_init, opt.scaling_lr_final, opt.lr_max_steps),
                   "rotation": expon_lr(opt.rotation_lr_init, opt.rotation_lr_final, opt.lr_max_steps)}
        self.optimizer = torch.optim.Adam([{"params": [self.p[n]], "lr": self.lr[n](0), "name": n} for n in self.NAMES],
                                          lr=0.0, eps=1e-15)
        self._reset_stats()

    @classmethod
    def from_tensors(cls, opt, backend, raw, moments=None, steps=None, max_radii2D=None, grad_accum=None, denom=None, gen=None):
        """A model around given raw parameters / Adam state (tests: the reference's fixtures, tests/golden/train/)."""
        m = cls.__new__(cls)
        m.opt, m.be, m.gen = opt, backend, gen
        dev = backend.device
        m.lo, m.hi = opt.scale_min * 2.0, opt.scale_max * 2.0
        m.p = {n: raw[n].to(dev).clone().requires_grad_(True) for n in cls.NAMES}
        m.lr = {"xyz": expon_lr(opt.position_lr_init, opt.position_lr_final, opt.lr_max_steps),
                "density": expon_lr(opt.density_lr_init, opt.density_lr_final, opt.lr_max_steps),
                "scaling": expon_lr(opt.scaling_lr_init, opt.scaling_lr_final, opt.lr_max_steps),
                "rotation": expon_lr(opt.rotation_lr_init, opt.rotation_lr_final, opt.lr_max_steps)}
        m.optimizer = torch.optim.Adam([{"params": [m.p[n]], "lr": m.lr[n](0), "name": n} for n in cls.NAMES], lr=0.0, eps=1e-15)
        if moments is not None:
            for n in cls.NAMES:
                m.optimizer.state[m.p[n]] = {"step": torch.as_tensor(steps[n] if steps is not None else 1.0),
                                             "exp_avg": moments[n][0].to(dev).clone(), "exp_avg_sq": moments[n][1].to(dev).clone()}
        P = m.p["xyz"].shape[0]
        m.max_radii2D = max_radii2D.to(dev).clone() if max_radii2D is not None else torch.zeros(P, device=dev)
        m.grad_accum = grad_accum.to(dev).clone() if grad_accum is not None else torch.zeros((P, 1), device=dev)
        m.denom = denom.to(dev).clone() if denom is not None else torch.zeros((P, 1), device=dev)
        return m

    # activations (gaussian_model.py:38-64)
    @staticmethod
    def inv_softplus(x):
        return torch.log(torch.exp(x) - 1)

    def scaling_act(self, x):
        return torch.sigmoid(x) * (self.hi - self.lo) + self.lo

    def scaling_inv(self, x):
        y = torch.relu((x - self.lo) / (self.hi - self.lo))
        return torch.log(y / (1 - y))

    def activated(self):
        return (self.p["xyz"], F.softplus(self.p["density"]), self.scaling_act(self.p["scaling"]),
                F.normalize(self.p["rotation"]))

    @property
    def P(self):
        return self.p["xyz"].shape[0]

    def _reset_stats(self):
        dev = self.be.device
        self.grad_accum = torch.zeros((self.P, 1), device=dev)
        self.denom = torch.zeros((self.P, 1), device=dev)

    def update_lr(self, it):
        for g in self.optimizer.param_groups:
            g["lr"] = self.lr[g["name"]](it)

    # optimizer-state surgery (gaussian_model.py:335-403)
    def _replace(self, new):
        for g in self.optimizer.param_groups:
            old = g["params"][0]
            st = self.optimizer.state.pop(old, None)
            t = new[g["name"]](old, st)
            param = t[0].detach().clone().requires_grad_(True)
            g["params"][0] = param
            if st is not None:
                st["exp_avg"], st["exp_avg_sq"] = t[1], t[2]
                self.optimizer.state[param] = st
            self.p[g["name"]] = param

    def _append(self, ext):
        def mk(name):
            def f(old, st):
                e = ext[name]
                if st is None:
                    return (torch.cat((old.detach(), e), 0), None, None)
                return (torch.cat((old.detach(), e), 0), torch.cat((st["exp_avg"], torch.zeros_like(e)), 0),
                        torch.cat((st["exp_avg_sq"], torch.zeros_like(e)), 0))
            return f
        self._replace({n: mk(n) for n in self.NAMES})
        self._reset_stats()   # densification_postfix zeroes the statistics (gaussian_model.py:423-425)

    def _keep(self, keep):
        def f(old, st):
            if st is None:
                return (old.detach()[keep], None, None)
            return (old.detach()[keep], st["exp_avg"][keep], st["exp_avg_sq"][keep])
        self._replace({n: f for n in self.NAMES})
        self.grad_accum, self.denom, self.max_radii2D = self.grad_accum[keep], self.denom[keep], self.max_radii2D[keep]

    @staticmethod
    def _rotmat(q):
        q = q / q.norm(dim=1, keepdim=True)
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)

    @torch.no_grad()
    def densify_and_prune_fused(self, bbox, normals_full=None):
        """The same through r2_gaussian_amd.densify (csrc/densify_ops.hip): one classify + one emit pass on the device."""
        from r2_gaussian_amd import densify as D
        opt = self.opt
        if normals_full is None:
            normals_full = torch.randn((2, self.P, 3), generator=self.gen)
        new_p, self.max_radii2D, self.grad_accum, self.denom = D.densify_and_prune_optimizer(
            self.optimizer, self.max_radii2D, self.grad_accum, self.denom, normals_full, opt.densify_grad_threshold,
            opt.densify_scale_threshold * 2.0, opt.density_min_threshold, bbox, (self.lo, self.hi),
            do_densify=self.P < opt.max_num_gaussians, max_screen_size=opt.max_screen_size,
            max_scale=opt.max_scale * 2.0 if opt.max_scale else None)
        self.p = dict(new_p)

    @torch.no_grad()
    def densify_and_prune(self, bbox, normals_full=None):
        """gaussian_model.py:503-550.  normals_full ([2,P,3], test hook): the split samples of parent i are normals_full[:, i]
        instead of fresh draws."""
        opt = self.opt
        thr_scale = opt.densify_scale_threshold * 2.0
        grads = self.grad_accum / self.denom
        grads[grads.isnan()] = 0.0
        if self.P < opt.max_num_gaussians:
            # clone: small Gaussians with a large view-space gradient; BOTH copies get half the density (:474-501)
            _x, dens, scal, _r = self.activated()
            sel = (grads.norm(dim=-1) >= opt.densify_grad_threshold) & (scal.max(dim=1).values <= thr_scale)
            half = self.inv_softplus(dens[sel] * 0.5)
            ext = {"xyz": self.p["xyz"].detach()[sel], "density": half, "scaling": self.p["scaling"].detach()[sel],
                   "rotation": self.p["rotation"].detach()[sel]}
            self.p["density"].data[sel] = half
            new_r = self.max_radii2D[sel]
            self._append(ext)
            self.max_radii2D = torch.cat([self.max_radii2D, new_r])
            # split: large Gaussians -> 2 samples from N(0, scale) in the local frame, scale / 1.6, density / 2 (:430-472)
            n0 = self.P
            pad = torch.zeros(n0, device=grads.device)
            pad[:grads.shape[0]] = grads.squeeze(-1)
            _x, dens, scal, _r = self.activated()
            sel = (pad >= opt.densify_grad_threshold) & (scal.max(dim=1).values > thr_scale)
            stds = scal[sel].repeat(2, 1)
            if normals_full is not None:
                nf = normals_full.to(stds.device)
                so = sel[:nf.shape[1]]
                samples = torch.cat([nf[0][so], nf[1][so]], 0) * stds
            else:
                samples = (torch.randn(stds.shape, generator=self.gen) * stds.cpu()).to(stds.device)   # seeded CPU stream
            R = self._rotmat(self.p["rotation"].detach()[sel]).repeat(2, 1, 1)
            ext = {"xyz": torch.bmm(R, samples.unsqueeze(-1)).squeeze(-1) + self.p["xyz"].detach()[sel].repeat(2, 1),
                   "density": self.inv_softplus(dens[sel].repeat(2, 1) * 0.5),
                   "scaling": self.scaling_inv(scal[sel].repeat(2, 1) / 1.6),
                   "rotation": self.p["rotation"].detach()[sel].repeat(2, 1)}
            new_r = self.max_radii2D[sel].repeat(2)
            self._append(ext)
            self.max_radii2D = torch.cat([self.max_radii2D, new_r])
            keep = ~torch.cat([sel, torch.zeros(2 * int(sel.sum()), dtype=torch.bool, device=sel.device)])
            self._keep(keep)
        xyz, dens, scal, _r = self.activated()
        b = bbox.to(xyz.device)
        prune = (dens < opt.density_min_threshold).squeeze(-1) | ((xyz < b[0]) | (xyz > b[1])).any(dim=1)
        if opt.max_screen_size:                                  # gaussian_model.py:540-542
            prune = prune | (self.max_radii2D > opt.max_screen_size)
        if opt.max_scale:                                        # :543-545, threshold x volume_to_world (train.py:53)
            prune = prune | (scal.max(dim=1).values > opt.max_scale * 2.0)
        self._keep(~prune)


def ssim(img1, img2, window_size=11):
    """loss_utils.py:57-104: 11x11 Gaussian window (sigma 1.5), zero padding, mean of the SSIM map; img [1,H,W]."""
    g = torch.tensor([math.exp(-((x - window_size // 2) ** 2) / (2 * 1.5 ** 2)) for x in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    # The window lives at the START of a 4 KB buffer: the vendor convolution on this stack reads past the end of its
    # 484-byte weight tensor (a pure-torch loop faults when the window is the last block of an allocator segment,
    # scripts/miopen_overread_probe.py); the reference's own ssim() allocates it bare.
    buf = torch.zeros(1024, device=img1.device)
    buf[:window_size * window_size] = (g @ g.t()).float().reshape(-1).to(img1.device)
    w = buf[:window_size * window_size].view(1, 1, window_size, window_size)
    a, b = img1[None], img2[None]
    pad = window_size // 2
    mu1, mu2 = F.conv2d(a, w, padding=pad), F.conv2d(b, w, padding=pad)
    s1 = F.conv2d(a * a, w, padding=pad) - mu1 * mu1
    s2 = F.conv2d(b * b, w, padding=pad) - mu2 * mu2
    s12 = F.conv2d(a * b, w, padding=pad) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))).mean()


def tv3d_mean(vol):
    d0, d1, d2 = (torch.diff(vol, dim=k).abs().sum() for k in range(3))
    n = vol.shape
    return (d0 + d1 + d2) / ((n[0] - 1) * n[1] * n[2] + n[0] * (n[1] - 1) * n[2] + n[0] * n[1] * (n[2] - 1))


def psnr3d(case, model):
    with torch.no_grad():
        x, d, s, r = model.activated()
        vol = model.be.query(x, d, s, r, case.center, case.nVoxel, case.sVoxel)
        return S.psnr3d(case.vol_gt, vol.detach().cpu())


def train(case, opt, backend_name, eval_every=100, seed=0, log=None, fused_losses=False, fused_densify=False, views_per_step=1,
          data_parallel=False, return_model=False, views_per_rank=1, exchange="accum"):
    """-> dict(iters=[...], psnr=[...], P=[...], it_per_s=...).  train.py:97-177.
    fused_losses (hip backend): the loss stack through r2_gaussian_amd.losses (one autograd node each) instead of torch ops.
    fused_densify (hip backend): densification statistics + densify / prune through r2_gaussian_amd.densify.
    views_per_step = W > 1: W views per optimiser step, gradients averaged -- rendered one after the other by this process, or
    with data_parallel (torch.distributed initialised, world size W) ONE per rank with r2_gaussian_amd.dist doing the exchange:
    one all-reduce of the parameter gradients, sum / max reductions of the densification statistics.  Both
    orders of evaluation draw the same random stream (view order, TV centre, split samples), so every rank and the
    single-process run hold the same model.
    views_per_rank = V > 1: every rank renders V views per optimiser step (W = world x V views per step, ONE
    parameter update, no stale gradients).  exchange = "accum": the rank sums its V gradient blocks locally and the step costs
    one all-reduce (1/V exchanges per view, all of it exposed); "sync2": every view's block is all-reduced on its own, the
    reduction of view j running (async) behind the render of view j+1 -- only the last one is exposed.  Both are synchronous
    data parallelism; they differ in how the float sums associate, and the single-process run (data_parallel=False) reproduces
    either association with "virtual ranks", so the comparison can be bit for bit."""
    from r2_gaussian_amd import dist as D
    be = Backend(backend_name)
    gen = torch.Generator().manual_seed(seed)          # TV centres, split samples
    pyrng = random.Random(seed)                        # view order (train.py:104-106)
    model = Model(case, opt, be, gen)
    dev = be.device
    gts = [p.to(dev) for p in case.projs]
    tvN = torch.tensor([opt.tv_vol_size] * 3)
    tvS = case.dVoxel * tvN
    W = int(views_per_step)
    V = int(views_per_rank)
    assert W % V == 0 and exchange in ("accum", "sync2")
    n_ranks = W // V                                   # real ranks (data_parallel) or virtual ones (single process)
    if data_parallel:
        assert D.world() == n_ranks, (D.world(), n_ranks)
    out = {"iters": [0], "psnr": [psnr3d(case, model)], "P": [model.P], "backend": backend_name}
    stack = []
    t_train = 0.0
    for it in range(1, opt.iterations + 1):
        t0 = time.perf_counter()
        model.update_lr(it)
        step_views = []
        for _ in range(W):
            if not stack:
                stack = list(range(len(case.views)))
            step_views.append(stack.pop(pyrng.randint(0, len(stack) - 1)))
        # view of (rank r, slot j) = step_views[r * V + j]
        mine = step_views[D.rank() * V:(D.rank() + 1) * V] if data_parallel else step_views
        c = None
        if opt.lambda_tv > 0:   # one TV patch per optimiser step: the same centre for all of the step's views / ranks
            c = (case.bbox[0] + tvS / 2) + (case.bbox[1] - tvS - case.bbox[0]) * torch.rand(3, generator=gen)
        inc_gn = inc_dn = rad_max = None
        flats, stats_v = [], []                          # W > 1: one packed [P,11] gradient block + statistics per view
        for vi in mine:
            x, d, s, r = model.activated()
            pkg = be.render(case.views[vi], x, d, s, r)
            img = pkg["render"]
            if fused_losses:
                from r2_gaussian_amd import losses as FL
                loss, _parts = FL.image_loss(img, gts[vi], opt.lambda_dssim)
            else:
                loss = (img - gts[vi]).abs().mean()
                if opt.lambda_dssim > 0:
                    loss = loss + opt.lambda_dssim * (1.0 - ssim(img, gts[vi]))
            if c is not None:
                vol = be.query(x, d, s, r, c, tvN, tvS)
                loss = loss + opt.lambda_tv * (FL.tv_3d_loss(vol) if fused_losses else tv3d_mean(vol))
            loss.backward()   # a step's views accumulate into .grad
            with torch.no_grad():
                if W == 1 and fused_densify:
                    from r2_gaussian_amd import densify as FD
                    FD.densification_stats(pkg["radii"], pkg["viewspace_points"].grad, model.max_radii2D, model.grad_accum,
                                           model.denom)
                elif W == 1:
                    vis, radii = pkg["visibility_filter"].to(dev), pkg["radii"].to(dev)
                    model.max_radii2D[vis] = torch.max(model.max_radii2D[vis], radii[vis].float())
                    g2 = pkg["viewspace_points"].grad
                    model.grad_accum[vis] += g2[vis, :2].norm(dim=-1, keepdim=True)
                    model.denom[vis] += 1
                else:   # the step's statistics increments (train.py:151-154 per view), reduced over ranks below
                    vis, radii = pkg["visibility_filter"].to(dev), pkg["radii"].to(dev).float()
                    gn = torch.where(vis, pkg["viewspace_points"].grad[:, :2].norm(dim=-1), torch.zeros((), device=dev))
                    stats_v.append((gn, vis.float(), radii))
                    params = [model.p[n] for n in model.NAMES]
                    flat = D.pack_grads(*(p_.grad for p_ in params))   # this view's gradient block, on its own
                    for p_ in params:
                        p_.grad = None
                    if data_parallel and exchange == "sync2":
                        # its all-reduce starts NOW and runs behind the next view's render (waited for after the loop)
                        flats.append((flat, D.allreduce_grads(flat, average=False, async_op=True)))
                    else:
                        flats.append((flat, None))
        with torch.no_grad():
            if W > 1:
                params = [model.p[n] for n in model.NAMES]

                def fold(ts):
                    acc = ts[0]
                    for t_ in ts[1:]:
                        acc = acc + t_
                    return acc

                def by_rank(items):   # [[rank 0's V items], [rank 1's], ...] (single process: virtual ranks)
                    return [items[r * V:(r + 1) * V] for r in range(len(items) // V)]
                if data_parallel:
                    if exchange == "sync2":
                        for _f, h in flats:
                            if h is not None:
                                h.wait()
                        total = fold([f_ for f_, _h in flats])          # slot sums, added in slot order
                    else:
                        total = fold([f_ for f_, _h in flats])          # local sum in view order ...
                        D.allreduce_grads(total, average=False)         # ... ONE all-reduce of the [P,11] gradient block
                    inc_gn, inc_dn = fold([s_[0] for s_ in stats_v]), fold([s_[1] for s_ in stats_v])
                    rad_max = stats_v[0][2]
                    for s_ in stats_v[1:]:
                        rad_max = torch.max(rad_max, s_[2])
                    inc_gn, inc_dn, rad_max = D.allreduce_densify_stats(inc_gn, inc_dn, rad_max)
                else:   # the same sums with the same association, ranks emulated
                    fr = by_rank([f_ for f_, _h in flats])
                    if exchange == "sync2":
                        total = fold([fold([fr[r][j] for r in range(n_ranks)]) for j in range(V)])
                    else:
                        total = fold([fold(fr[r]) for r in range(n_ranks)])
                    sr = by_rank(stats_v)
                    inc_gn = fold([fold([s_[0] for s_ in sr[r]]) for r in range(n_ranks)])
                    inc_dn = fold([fold([s_[1] for s_ in sr[r]]) for r in range(n_ranks)])
                    rad_max = stats_v[0][2]
                    for s_ in stats_v[1:]:
                        rad_max = torch.max(rad_max, s_[2])
                for p_, g_ in zip(params, D.unpack_grads(total)):
                    p_.grad = (g_.reshape(p_.shape) / W).contiguous()
                model.max_radii2D = torch.max(model.max_radii2D, rad_max)
                model.grad_accum += inc_gn[:, None]
                model.denom += inc_dn[:, None]
            if it < opt.densify_until_iter and it > opt.densify_from_iter and it % opt.densification_interval == 0:
                if fused_densify:
                    model.densify_and_prune_fused(case.bbox)
                else:
                    model.densify_and_prune(case.bbox)
                if data_parallel:   # every rank took the same decisions (SURVEY.md 8e "consistency")
                    for n in model.NAMES:
                        D.assert_replicas_equal(model.p[n].detach(), n)
            if model.P == 0:
                raise ValueError("No Gaussian left")
            if it < opt.iterations:
                model.optimizer.step()
                model.optimizer.zero_grad(set_to_none=True)
        if be.name == "hip":
            torch.cuda.synchronize()
        t_train += time.perf_counter() - t0
        if it % eval_every == 0 or it == opt.iterations:
            out["iters"].append(it)
            out["psnr"].append(psnr3d(case, model))
            out["P"].append(model.P)
            if log:
                log("it %5d  P %6d  psnr3d %.3f dB  loss %.4e  (%.1f it/s)" % (it, model.P, out["psnr"][-1], float(loss.detach()),
                                                                              it / t_train))
    out["it_per_s"] = opt.iterations / t_train
    if return_model:
        out["model"] = model
    return out
