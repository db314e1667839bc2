"""Writes the bench.py workload (seeded cloud + views + pixel gradient) as one flat binary for scripts/cbench.cpp.

    python scripts/dump_scene.py [P | small | large] [HW] [views] [name]  ->  scripts/_scene/<name>.bin   (git-ignored; travels with gpurun)

Layout (little endian): int32 {P, V, H, W}; f32 means3D[P,3], density[P], scales[P,3], rotations[P,4];
per view: f32 viewmatrix[16], projmatrix[16], campos[3], tanfovx, tanfovy, int32 mode; f32 dL[H*W]."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from r2_gaussian_amd import scene as S  # noqa: E402

# [P | small | large] [HW] [views] [name]: a recipe name instead of P = a TRAINED cloud (tests/trained_cloud.py: looked up or trained on
# the spot, needs the GPU); name = the file's stem under scripts/_scene/ (scripts/cbench reads $R2_SCENE, default scene.bin)
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 512
V = int(sys.argv[3]) if len(sys.argv) > 3 else 50
NAME = sys.argv[4] if len(sys.argv) > 4 else "scene"
if len(sys.argv) > 1 and not sys.argv[1].isdigit():
    from tests import trained_cloud as TCl
    cloud, _info = TCl.load(sys.argv[1])
    P = cloud.xyz.shape[0]
else:
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    cloud = S.make_cloud(P, seed=0)
views = S.make_views(V, (HW, HW))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_scene", NAME + ".bin")
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, "wb") as f:
    np.array([P, V, HW, HW], dtype=np.int32).tofile(f)
    for t in (cloud.xyz, cloud.density, cloud.scales, cloud.rotations):
        t.detach().contiguous().numpy().astype(np.float32).tofile(f)
    for v in views:
        v.world_view_transform.contiguous().numpy().astype(np.float32).tofile(f)
        v.full_proj_transform.contiguous().numpy().astype(np.float32).tofile(f)
        v.camera_center.contiguous().numpy().astype(np.float32).tofile(f)
        np.array([v.tanfovx, v.tanfovy], dtype=np.float32).tofile(f)
        np.array([v.mode], dtype=np.int32).tofile(f)
    S.make_pixel_grad(HW, HW).contiguous().numpy().astype(np.float32).tofile(f)
print(out, os.path.getsize(out))
