"""cProfile of the bench step loop: where does the HOST time of a training view go?"""
import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from r2_gaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, scene as S
dev = torch.device("cuda:0")
P, HW = 300000, 512
c = S.make_cloud(P, seed=0); views = S.make_views(50, (HW, HW))
xyz, dens, scal, rot = (t.to(dev).requires_grad_(True) for t in (c.xyz, c.density, c.scales, c.rotations))
dL = S.make_pixel_grad(HW, HW).to(dev)
rs = [GaussianRasterizer(GaussianRasterizationSettings(HW, HW, v.tanfovx, v.tanfovy, 1.0, v.world_view_transform.to(dev), v.full_proj_transform.to(dev), v.camera_center.to(dev), False, v.mode, False)) for v in views]
def step(k):
    m2 = torch.zeros_like(xyz, requires_grad=True)
    img, radii = rs[k % 50](means3D=xyz, means2D=m2, opacities=dens, scales=scal, rotations=rot)
    for p in (xyz, dens, scal, rot): p.grad = None
    img.backward(dL)
for k in range(60): step(k)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for k in range(300): step(k)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:5000])
