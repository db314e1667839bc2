#!/bin/bash
# round 6: A/B of library variants through the C host, two alternating repetitions; LIBS="[ENV=..:]lib.so ..." [TESTS="pytest args"]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r6d}
mkdir -p gpurun_out/ab
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest $TESTS -q -m gpu --durations=5 2>&1 | tail -${TAIL:-12} | tee gpurun_out/pytest_$TAG.log; fi
for rep in 1 2; do
  for L in $LIBS; do
    E=""; F=$L; case $L in *:*) E=${L%%:*}; F=${L#*:};; esac
    env $E timeout 300 scripts/cbench ${STEPS:-300} r2_gaussian_amd/$F ${SECTIONS:-single,stages} > gpurun_out/ab/${TAG}_${L}_$rep.txt 2>&1
    echo "== $L (rep $rep): $(grep -E 'BEST|raster\.|^voxel' gpurun_out/ab/${TAG}_${L}_$rep.txt | tr '\n' ';' | sed 's/  */ /g' | sed 's/raster\.//g')"
  done
done | tee gpurun_out/ab/${TAG}_summary.txt
if [ -n "$TSLIB" ]; then
  timeout 200 scripts/cbench 100 r2_gaussian_amd/$TSLIB single > gpurun_out/ab/${TAG}_ts.txt 2>&1
  echo "== stamps $TSLIB"; grep -E "BEST|TS |TSD " gpurun_out/ab/${TAG}_ts.txt
fi
