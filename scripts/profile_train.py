"""Where does a training iteration spend its time?  Coarse host-side breakdown (synchronising after every phase) of the
miniature trainer on the HIP backend at 512^2 / 256^3 / 90k Gaussians.  gpurun_out/profile_train.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests import mini_trainer as T

case = T.Case(detector=512, n_vol=256, n_views=50, p_gt=20000, n_init=int(os.environ.get('N_INIT', '90000')), seed=2)
opt = T.Opt(iterations=400, densify_from_iter=10**9, densify_until_iter=0)
be = T.Backend("hip")
gen = torch.Generator().manual_seed(0)
model = T.Model(case, opt, be, gen)
dev = be.device
gts = [p.to(dev) for p in case.projs]
tvN = torch.tensor([32] * 3); tvS = case.dVoxel * tvN
acc = {}
def tick(name, t0):
    torch.cuda.synchronize(); t = time.perf_counter()
    if os.environ.get('TRACE') == '1':
        print(it, name, flush=True)
    acc[name] = acc.get(name, 0.0) + t - t0
    return t
for it in range(1, 301):
    if it == 101:
        acc.clear()
    torch.cuda.synchronize(); t = time.perf_counter()
    model.update_lr(it)
    x, d, s, r = model.activated(); t = tick("activations", t)
    pkg = be.render(case.views[it % 50], x, d, s, r); img = pkg["render"]; t = tick("render_fwd", t)
    loss = (img - gts[it % 50]).abs().mean()
    if os.environ.get('NO_SSIM') != '1':
        loss = loss + 0.25 * (1.0 - T.ssim(img, gts[it % 50]))
    t = tick("l1+ssim_fwd", t)
    c = (case.bbox[0] + tvS / 2) + (case.bbox[1] - tvS - case.bbox[0]) * torch.rand(3, generator=gen)
    if os.environ.get('NO_TV') != '1':
        vol = be.query(x, d, s, r, c, tvN, tvS); t = tick("tv_query_fwd", t)
        loss = loss + 0.05 * T.tv3d_mean(vol); t = tick("tv_loss_fwd", t)
    loss.backward(); t = tick("backward_all", t)
    with torch.no_grad():
        if os.environ.get('NO_STATS') != '1':
            vis, radii = pkg["visibility_filter"], pkg["radii"]
            model.max_radii2D[vis] = torch.max(model.max_radii2D[vis], radii[vis].float())
            g2 = pkg["viewspace_points"].grad
            model.grad_accum[vis] += g2[vis, :2].norm(dim=-1, keepdim=True)
            model.denom[vis] += 1; t = tick("densify_stats", t)
        if os.environ.get('NO_ADAM') != '1':
            model.optimizer.step()
        model.optimizer.zero_grad(set_to_none=True); t = tick("adam", t)
n = 200
out = {k: round(1e6 * v / n, 1) for k, v in acc.items()}
out["total_us"] = round(sum(out.values()), 1)
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "profile_train.json"), "w"), indent=1)
