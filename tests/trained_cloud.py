"""Trained, densified clouds for the parity tests and bench.py --cloud (BASELINE configs[2] "full densification
to ~300k Gaussians").  The tensors are not committed (4-15 MB each): a cloud is looked up

  1. under $R2_CLOUD_DIR/<name>/ or gpurun_out/clouds/<name>/ (what scripts/train_cloud.py wrote),
  2. in the per-machine cache /tmp/r2_clouds/<name>/,
  3. and otherwise TRAINED on the spot with the HIP trainer (scripts/train_cloud.py recipes: ~10 s on an MI355X for "small"
     50k -> 92k, ~12 s for "large" 50k -> 335k; needs a GPU) and cached,

always as a point_cloud.pickle in the reference's model layout (gaussian_model.py:263-318) read back through
r2_gaussian_amd.model_io -- so the loader path is exercised on the way.  TEST INFRASTRUCTURE (trains with tests/mini_trainer.py).
"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
CACHE = "/tmp/r2_clouds"


def _find(name):
    for base in (os.environ.get("R2_CLOUD_DIR"), os.path.join(ROOT, "gpurun_out", "clouds"), CACHE):
        if not base:
            continue
        hits = sorted(glob.glob(os.path.join(base, name, "point_cloud", "iteration_*", "point_cloud.pickle")))
        if hits:
            return hits[-1]
    return None


def path(name, train=True):
    """Path of the cloud's point_cloud.pickle, training it first when it exists nowhere (needs cuda:0)."""
    p = _find(name)
    if p is None and train:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import train_cloud as TC
        _act, raw, info = TC.train_recipe(name)
        p = TC.save(CACHE, name, raw, info)
    return p


def load(name_or_path, train=True):
    """-> (scene.Cloud of CPU tensors (activated: what render() / query() feed the kernels), info dict)."""
    import torch
    from r2_gaussian_amd import model_io, scene as S
    p = name_or_path if os.path.isfile(str(name_or_path)) else path(name_or_path, train)
    if p is None:
        return None, None
    m = model_io.load_point_cloud(p, device="cpu")
    with torch.no_grad():
        x, d, s, r = (t.float().contiguous() for t in model_io.activate(m))
    info = {"path": p, "P": int(x.shape[0])}
    ij = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(p))), "info.json")
    if os.path.exists(ij):
        info.update(json.load(open(ij)))
    return S.Cloud(x, s, r, d.reshape(-1, 1)), info
