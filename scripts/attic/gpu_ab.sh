#!/bin/bash
# A/B of library builds through the C harness (single-view step + per-stage HIP-event times), same box, interleaved:
#   gpurun -- bash scripts/gpu_ab.sh "libA.so libB.so ..." [pytest-args]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
LIBS=${1:-"libr2hip_base.so libr2hip.so"}
shift
mkdir -p gpurun_out/ab
if [ -n "$1" ]; then
  timeout 1500 python -m pytest "$@" 2>&1 | tail -15
fi
for rep in 1 2; do
  for L in $LIBS; do
    echo "=== $L (rep $rep)"
    # "VAR=value:lib.so" runs the library with that environment variable set
    E=""; F=$L; case $L in *:*) E=${L%%:*}; F=${L#*:};; esac
    env $E timeout 200 scripts/cbench ${STEPS:-300} r2_gaussian_amd/$F single,stages > gpurun_out/ab/${L}_$rep.txt 2>&1
    grep -E "BEST|raster\." gpurun_out/ab/${L}_$rep.txt | tr '\n' ';' | sed 's/  */ /g'; echo
  done
done
