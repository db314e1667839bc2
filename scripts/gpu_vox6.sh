#!/bin/bash
# Voxelizer A/B of library builds on ONE box, alternating, through the C host: the 256^3 query of the benchmark cloud and of the two
# trained clouds (+ the 32^3 TV patch), after the voxel test files.   LIBS="libA.so libB.so" TESTS=1 bash scripts/gpu_vox6.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-vox6}
mkdir -p gpurun_out/ab scripts/_scene
if [ "${TESTS:-1}" = 1 ]; then
  timeout 1500 python -m pytest tests/test_voxel_gpu.py tests/test_voxel_sticks_gpu.py tests/test_variants_gpu.py tests/test_dispatch_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee gpurun_out/pytest_$TAG.log
fi
[ -f scripts/_scene/scene.bin ] || python scripts/dump_scene.py 300000 512 50 scene > /dev/null
[ -f scripts/_scene/trained_small.bin ] || python scripts/dump_scene.py small 512 50 trained_small 2>&1 | tail -1
[ -f scripts/_scene/trained_large.bin ] || python scripts/dump_scene.py large 512 50 trained_large 2>&1 | tail -1
for rep in 1 2; do
  for S in scene trained_small trained_large; do
    for L in ${LIBS:-libr2hip_prev.so libr2hip.so}; do
      E=""; F=$L; case $L in *:*) E=${L%%:*}; F=${L#*:};; esac
      env $E R2_SCENE=scripts/_scene/$S.bin timeout 300 scripts/cbench 20 r2_gaussian_amd/$F voxel > gpurun_out/ab/${TAG}_${S}_${L}_$rep.txt 2>&1
      echo "== $S $L (rep $rep): $(grep -E '^voxel|  voxel\.' gpurun_out/ab/${TAG}_${S}_${L}_$rep.txt | tr -s ' ' | tr '\n' ';')"
    done
  done
done | tee gpurun_out/ab/${TAG}_summary.txt
