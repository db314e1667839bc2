#!/bin/bash
# The per-round A/B on ALL rasterizer workloads, through the C host, on ONE box, alternating runs (a change tuned at
# 300k must not regress the others silently): headline 300k/512^2, B 50k/512^2, C 300k/560^2, E 1M/1024^2, trained 92k and 331k
# clouds (512^2).  LIBS="[ENV=..:]lib.so ..." ; prints views/s + stage times per (workload, library).
#   gpurun -- env TAG=r06x LIBS="libr2hip_base.so libr2hip.so" bash scripts/gpu_ab7.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-ab7}
mkdir -p gpurun_out/ab scripts/_scene
[ -f scripts/_scene/scene.bin ] || python scripts/dump_scene.py 300000 512 50 scene > /dev/null
python scripts/dump_scene.py 50000 512 50 B > /dev/null
python scripts/dump_scene.py 300000 560 50 C > /dev/null
python scripts/dump_scene.py 1000000 1024 60 E > /dev/null
python scripts/dump_scene.py small 512 50 trained_small 2>&1 | tail -1
python scripts/dump_scene.py large 512 50 trained_large 2>&1 | tail -1
for rep in 1 2; do
  for S in scene B C E trained_small trained_large; do
    for L in $LIBS; do
      E=""; F=$L; case $L in *:*) E=${L%%:*}; F=${L#*:};; esac
      ST=300; [ $S = E ] && ST=120
      env $E R2_SCENE=scripts/_scene/$S.bin timeout 300 scripts/cbench $ST r2_gaussian_amd/$F single,stages > gpurun_out/ab/${TAG}_${S}_${L}_$rep.txt 2>&1
      echo "== $S $L (rep $rep): $(grep -E 'BEST|raster\.' gpurun_out/ab/${TAG}_${S}_${L}_$rep.txt | tr '\n' ';' | sed 's/  */ /g' | sed 's/raster\.//g')"
    done
  done
done | tee gpurun_out/ab/${TAG}_summary.txt
