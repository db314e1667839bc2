#!/bin/bash
# same-box A/B of library builds on the 256^3 query through the C host: bash scripts/gpu_sticks_libs.sh libA.so libB.so ...
[ -f scripts/_scene/scene.bin ] || python scripts/dump_scene.py > /dev/null 2>&1
for rep in 1 2 3; do for L in "$@"; do
  echo -n "$L rep $rep: "; timeout 200 scripts/cbench 20 r2_gaussian_amd/$L voxel 2>&1 | grep -E "^voxel 256|  voxel\.(duplicate|sort|preprocess|scan)" | tr -s ' ' | tr '\n' ';'; echo
done; done
