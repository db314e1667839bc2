"""Where does the host time of one training view go?  Manual timers around the Python layers of forward and backward."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from r2_gaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, scene as S, _C, _lib
T = {}
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[name] = T.get(name, 0.0) + time.perf_counter() - t0; return r
    setattr(mod, name, g)
wrap(_C, "rasterize_gaussians"); wrap(_C, "rasterize_gaussians_backward")
L = _lib.lib()
class Timed:
    def __init__(self, fn, name): self.fn, self.name = fn, name
    def __call__(self, *a):
        t0 = time.perf_counter(); r = self.fn(*a); T[self.name] = T.get(self.name, 0.0) + time.perf_counter() - t0; return r
class TimedGap(Timed):
    """also: time from the forward C call's return to the backward C call's entry, and to its return"""
    last_fwd_end = 0.0
    def __call__(self, *a):
        t0 = time.perf_counter()
        if self.name == "C.backward":
            T["gap fwd-return -> bwd-entry"] = T.get("gap fwd-return -> bwd-entry", 0.0) + t0 - TimedGap.last_fwd_end
        r = self.fn(*a)
        t1 = time.perf_counter()
        T[self.name] = T.get(self.name, 0.0) + t1 - t0
        if self.name == "C.forward":
            TimedGap.last_fwd_end = t1
        return r
L.r2_raster_forward = TimedGap(L.r2_raster_forward, "C.forward"); L.r2_raster_backward = TimedGap(L.r2_raster_backward, "C.backward")
dev = torch.device("cuda:0")
P, HW = 300000, 512
c = S.make_cloud(P, seed=0); views = S.make_views(50, (HW, HW))
xyz, dens, scal, rot = (t.to(dev).requires_grad_(True) for t in (c.xyz, c.density, c.scales, c.rotations))
dL = S.make_pixel_grad(HW, HW).to(dev)
rs = [GaussianRasterizer(GaussianRasterizationSettings(HW, HW, v.tanfovx, v.tanfovy, 1.0, v.world_view_transform.to(dev), v.full_proj_transform.to(dev), v.camera_center.to(dev), False, v.mode, False)) for v in views]
tf = tb = 0.0
def step(k):
    global tf, tb
    t0 = time.perf_counter()
    m2 = torch.zeros_like(xyz, requires_grad=True)
    img, radii = rs[k % 50](means3D=xyz, means2D=m2, opacities=dens, scales=scal, rotations=rot)
    for p in (xyz, dens, scal, rot): p.grad = None
    t1 = time.perf_counter()
    img.backward(dL)
    t2 = time.perf_counter(); tf += t1 - t0; tb += t2 - t1
for k in range(60): step(k)
torch.cuda.synchronize(); T.clear(); tf = tb = 0.0; _lib.sync_wait_stats(True); _lib.profile_host(True)
N = 300
t0 = time.perf_counter()
for k in range(N): step(k)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
w, n = _lib.sync_wait_stats(True)
pre, post, nf = _lib.profile_host(True)
print("C forward: %.1f us before the wait, %.1f us after it | gap forward-return -> backward-entry %.1f us" % (pre, post, 1e6 * T["gap fwd-return -> bwd-entry"] / N))
print("step %.1f us | forward part %.1f (python _C.rasterize %.1f, C call %.1f of which wait %.1f) | backward part %.1f (python _C.backward %.1f, C call %.1f)" % (
    1e6 * dt / N, 1e6 * tf / N, 1e6 * T["rasterize_gaussians"] / N, 1e6 * T["C.forward"] / N, w / n, 1e6 * tb / N, 1e6 * T["rasterize_gaussians_backward"] / N, 1e6 * T["C.backward"] / N))
