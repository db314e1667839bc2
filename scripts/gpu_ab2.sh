#!/bin/bash
# A/B of library builds through the C harness (single-view step + per-stage HIP-event times) + in-kernel stamps of one of them
#   gpurun -- bash scripts/gpu_ab2.sh "libA.so libB.so ..." [stamped.so] [pytest args]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
LIBS=${1:-"libr2hip_base.so libr2hip.so"}
TS=${2:-}
shift; shift
mkdir -p gpurun_out/ab gpurun_out/ts
if [ -n "$1" ]; then timeout 1200 python -m pytest "$@" 2>&1 | tail -8; fi
for rep in 1 2; do
  for L in $LIBS; do
    E=""; F=$L; case $L in *:*) E=${L%%:*}; F=${L#*:};; esac
    env $E timeout 200 scripts/cbench ${STEPS:-300} r2_gaussian_amd/$F single,stages > gpurun_out/ab/${L}_$rep.txt 2>&1
    echo "== $L (rep $rep): $(grep -E 'BEST|raster\.' gpurun_out/ab/${L}_$rep.txt | tr '\n' ';' | sed 's/  */ /g')"
  done
done
if [ -n "$TS" ]; then
  timeout 200 scripts/cbench 100 r2_gaussian_amd/$TS single > gpurun_out/ts/$TS.txt 2>&1
  echo "== stamps $TS"; grep -E "BEST|TS " gpurun_out/ts/$TS.txt
fi
