"""tests/golden/train/: fixtures produced by the REFERENCE's own Python for the rows either side of the hot path
(SURVEY.md 8f-1 / 8f-2), so that the repo's restatement (tests/mini_trainer.py) and the HIP operators
(r2_gaussian_amd/losses.py, densify.py) are pinned against the reference and not against each other:

  losses.npz    l1_loss / ssim / tv_3d_loss values AND autograd gradients   r2_gaussian/utils/loss_utils.py:19-104
  lr.npz        get_expon_lr_func at a spread of steps, the four groups      r2_gaussian/utils/gaussian_utils.py:13-46
                + training_setup's group order / names / initial lr / eps    r2_gaussian/gaussian/gaussian_model.py:188-235
  densify_*.npz GaussianModel.densify_and_prune (clone + split + prune + Adam-state surgery), inputs and outputs,
                with the split's normal samples recorded                    r2_gaussian/gaussian/gaussian_model.py:380-550
                A default thresholds | B max_screen_size + max_scale set | C prune only (max_num_gaussians reached)

Needs /root/reference (build container only).  The reference's GaussianModel hard-codes device="cuda" in the tensors it
creates (gaussian_model.py:425-426,433,443,467; gaussian_utils.py:56); there is no CUDA device here, so the name `torch` inside
those two modules is replaced by a proxy that forwards everything to torch but drops device="cuda" -- and whose normal()
draws the unit samples from a seeded generator, RECORDS them and returns mean + std * z (what torch.normal(mean, std)
computes).  Nothing else of the reference is touched; third-party imports its module chain never uses here are stubbed as
in make_golden_io.py.

    python tests/golden/make_golden_train.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
OUT = os.path.join(HERE, "train")


class TorchOnCPU:
    """Stands in for the module `torch` inside the reference's gaussian_model / gaussian_utils."""

    def __init__(self):
        self.gen = torch.Generator().manual_seed(1234)
        self.normal_samples = []

    def __getattr__(self, name):
        return getattr(torch, name)

    @staticmethod
    def _cpu(kw):
        if str(kw.get("device", "")).startswith("cuda"):
            kw = dict(kw, device="cpu")
        return kw

    def zeros(self, *a, **kw):
        return torch.zeros(*a, **self._cpu(kw))

    def ones(self, *a, **kw):
        return torch.ones(*a, **self._cpu(kw))

    def empty(self, *a, **kw):
        return torch.empty(*a, **self._cpu(kw))

    def tensor(self, *a, **kw):
        return torch.tensor(*a, **self._cpu(kw))

    def normal(self, mean, std):
        z = torch.randn(std.shape, generator=self.gen)
        self.normal_samples.append(z)
        return mean + std * z


def import_reference():
    for name, attrs in (("plyfile", ("PlyData", "PlyElement")), ("simple_knn", ()), ("simple_knn._C", ("distCUDA2",))):
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, None)
        sys.modules[name] = m
    sys.path.insert(0, REF)
    import r2_gaussian.gaussian.gaussian_model as gm
    import r2_gaussian.utils.gaussian_utils as gu
    import r2_gaussian.utils.loss_utils as lu
    proxy = TorchOnCPU()
    gm.torch = proxy
    gu.torch = proxy
    return gm, gu, lu, proxy


def training_args(**kw):
    a = types.SimpleNamespace(position_lr_init=0.0002, position_lr_final=0.00002, position_lr_max_steps=30000,
                              density_lr_init=0.01, density_lr_final=0.001, density_lr_max_steps=30000,
                              scaling_lr_init=0.005, scaling_lr_final=0.0005, scaling_lr_max_steps=30000,
                              rotation_lr_init=0.001, rotation_lr_final=0.0001, rotation_lr_max_steps=30000)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


SCALE_BOUND = np.array([0.0005, 0.5]) * 2.0     # dataset.scale_min / scale_max x volume_to_world (train.py:59-61)
BBOX = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])


def densify_scene(gm, P=1500, seed=5):
    """A model that takes every branch: clone, split, prune by density, by box, by screen size, by world size."""
    g = torch.Generator().manual_seed(seed)
    m = gm.GaussianModel(SCALE_BOUND)
    lo, hi = float(SCALE_BOUND[0]), float(SCALE_BOUND[1])
    xyz = (torch.rand(P, 3, generator=g) * 2 - 1) * 0.8
    xyz[::23] = torch.tensor([0.995, 0.0, 0.0])                # splits near the box face: some children land outside
    xyz[::31, 1] = 1.2                                         # outside the box
    scal = torch.exp(torch.rand(P, 3, generator=g) * 5.0 - 6.0).clamp(lo * 1.01, hi * 0.99)
    dens = torch.rand(P, 1, generator=g) * 0.3
    dens[::17] = 1.5e-5                                        # falls below the 1e-5 threshold when halved
    dens[::29] = 5e-6                                          # pruned outright
    m._xyz = torch.nn.Parameter(xyz.clone().requires_grad_(True))
    m._density = torch.nn.Parameter(m.density_inverse_activation(dens).requires_grad_(True))
    m._scaling = torch.nn.Parameter(m.scaling_inverse_activation(scal).requires_grad_(True))
    m._rotation = torch.nn.Parameter(torch.randn(P, 4, generator=g).requires_grad_(True))
    m.max_radii2D = torch.rand(P, generator=g) * 50
    m.spatial_lr_scale = 1.0
    m.training_setup(training_args())
    # one optimizer step so that exp_avg / exp_avg_sq / step exist
    for p in (m._xyz, m._density, m._scaling, m._rotation):
        p.grad = torch.randn(p.shape, generator=g) * 1e-3
    m.update_learning_rate(1)
    m.optimizer.step()
    m.optimizer.zero_grad(set_to_none=True)
    m.xyz_gradient_accum = torch.rand(P, 1, generator=g) * 3e-4
    m.denom = torch.randint(0, 4, (P, 1), generator=g).float()   # some zeros: 0/0 -> nan -> 0
    return m


def snapshot(m, prefix):
    out = {prefix + "xyz": m._xyz, prefix + "density": m._density, prefix + "scaling": m._scaling, prefix + "rotation": m._rotation,
           prefix + "max_radii2D": m.max_radii2D, prefix + "grad_accum": m.xyz_gradient_accum, prefix + "denom": m.denom}
    for grp in m.optimizer.param_groups:
        st = m.optimizer.state[grp["params"][0]]
        out[prefix + grp["name"] + ".m"] = st["exp_avg"]
        out[prefix + grp["name"] + ".v"] = st["exp_avg_sq"]
        out[prefix + grp["name"] + ".step"] = torch.as_tensor(st["step"])
    return {k: v.detach().cpu().numpy().copy() for k, v in out.items()}


def run_densify(gm, proxy, name, max_grad, min_density, max_screen_size, max_scale, max_num_gaussians, scale_thr):
    m = densify_scene(gm)
    P = m._xyz.shape[0]
    data = snapshot(m, "in.")
    proxy.normal_samples.clear()
    with torch.no_grad():   # train.py:149
        m.densify_and_prune(max_grad, min_density, max_screen_size, max_scale, max_num_gaussians, scale_thr, BBOX)
    data.update(snapshot(m, "out."))
    # the split's unit samples, per parent row: normals[c, i] = sample of child c of parent i (gaussian_model.py:441-447:
    # stds = scaling[selected].repeat(2, 1) -> first all first children, then all second children)
    normals = np.zeros((2, P, 3), np.float32)
    if proxy.normal_samples:
        z = proxy.normal_samples[0].numpy()
        # recompute the split selection exactly as the reference did: over the rows that existed before the clone step
        ref = densify_scene(gm)
        grads = ref.xyz_gradient_accum / ref.denom
        grads[grads.isnan()] = 0.0
        sel = (grads.squeeze(-1) >= max_grad) & (ref.get_scaling.max(dim=1).values > scale_thr)
        idx = torch.nonzero(sel).squeeze(-1).numpy()
        assert z.shape[0] == 2 * len(idx), (z.shape, len(idx))
        normals[0, idx] = z[:len(idx)]
        normals[1, idx] = z[len(idx):]
    data["normals"] = normals
    data["cfg"] = np.array([max_grad, min_density, max_screen_size or 0.0, max_scale or 0.0, max_num_gaussians, scale_thr],
                           np.float64)   # 0 = None for the two optional thresholds
    data["scale_bound"] = SCALE_BOUND
    data["bbox"] = BBOX.numpy()
    np.savez_compressed(os.path.join(OUT, "densify_%s.npz" % name), **data)
    print("densify_%s: P %d -> %d, split samples %d" % (name, P, m._xyz.shape[0], 0 if not proxy.normal_samples else len(proxy.normal_samples[0])))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    gm, gu, lu, proxy = import_reference()

    # ---- losses: values and autograd gradients
    g = torch.Generator().manual_seed(21)
    data = {}
    for i, (H, W) in enumerate(((32, 32), (50, 70), (96, 64))):
        gt = torch.rand(1, H, W, generator=g)
        img = (gt + 0.1 * torch.randn(1, H, W, generator=g)).clamp_min(0).requires_grad_(True)
        l1 = lu.l1_loss(img, gt)
        (gl1,) = torch.autograd.grad(l1, img)
        ss = lu.ssim(img, gt)
        (gss,) = torch.autograd.grad(ss, img)
        data.update({"img%d" % i: img.detach().numpy(), "gt%d" % i: gt.numpy(), "l1_%d" % i: l1.detach().numpy(),
                     "l1_grad%d" % i: gl1.numpy(), "ssim_%d" % i: ss.detach().numpy(), "ssim_grad%d" % i: gss.numpy()})
    for i, shape in enumerate(((32, 32, 32), (12, 10, 14))):
        vol = torch.rand(*shape, generator=g).requires_grad_(True)
        tv = lu.tv_3d_loss(vol, reduction="mean")
        (gtv,) = torch.autograd.grad(tv, vol)
        data.update({"vol%d" % i: vol.detach().numpy(), "tv_%d" % i: tv.detach().numpy(), "tv_grad%d" % i: gtv.numpy(),
                     "tv_sum_%d" % i: lu.tv_3d_loss(vol.detach(), reduction="sum").numpy()})
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **data)

    # ---- learning-rate schedules + optimizer groups
    steps = np.array([0, 1, 2, 100, 499, 5000, 14999, 29999, 30000, 40000])
    a = training_args()
    m = densify_scene(gm, P=8)
    lr = {}
    for name, f in (("xyz", m.xyz_scheduler_args), ("density", m.density_scheduler_args), ("scaling", m.scaling_scheduler_args),
                    ("rotation", m.rotation_scheduler_args)):
        lr[name] = np.array([f(int(s)) for s in steps], np.float64)
    fresh = gm.GaussianModel(SCALE_BOUND)
    fresh._xyz, fresh._density, fresh._scaling, fresh._rotation = m._xyz, m._density, m._scaling, m._rotation
    fresh.spatial_lr_scale = 1.0
    fresh.training_setup(a)
    groups = fresh.optimizer.param_groups
    np.savez_compressed(os.path.join(OUT, "lr.npz"), steps=steps, group_names=np.array([g_["name"] for g_ in groups]),
                        group_lr0=np.array([g_["lr"] for g_ in groups]), eps=np.array([g_["eps"] for g_ in groups]),
                        betas=np.array([g_["betas"] for g_ in groups]),
                        disabled=np.array([gu.get_expon_lr_func(0.0, 0.0, max_steps=10)(3), gu.get_expon_lr_func(1e-3, 1e-4, max_steps=10)(-1)]),
                        **{"lr_" + k: v for k, v in lr.items()})

    # ---- densify_and_prune (train.py:155-168 passes thresholds scaled by volume_to_world = 2)
    run_densify(gm, proxy, "A", 1.5e-4, 1e-5, None, None, 500000, 0.02 * 2.0)
    run_densify(gm, proxy, "B", 1.5e-4, 1e-5, 35.0, 0.15 * 2.0, 500000, 0.02 * 2.0)
    run_densify(gm, proxy, "C", 1.5e-4, 1e-5, 35.0, 0.15 * 2.0, 10, 0.02 * 2.0)
    print("wrote", sorted(os.listdir(OUT)), sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT)) // 1024, "KiB")
