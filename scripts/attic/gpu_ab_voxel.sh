#!/bin/bash
# A/B of library builds / environment switches on the voxelizer sections of the C harness (256^3 query + 32^3 TV patch), same box:
#   gpurun -- bash scripts/gpu_ab_voxel.sh "ENV=1:libr2hip.so libr2hip.so" [pytest-args]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
LIBS=${1:-"libr2hip.so"}
shift
mkdir -p gpurun_out/ab
if [ -n "$1" ]; then
  timeout 1500 python -m pytest "$@" 2>&1 | tail -15
fi
for rep in 1 2; do
  for L in $LIBS; do
    echo "=== $L (rep $rep)"
    E=""; F=$L; case $L in *:*) E=${L%%:*}; F=${L#*:};; esac
    env $E timeout 200 scripts/cbench ${STEPS:-100} r2_gaussian_amd/$F voxel > gpurun_out/ab/vox_${L}_$rep.txt 2>&1
    grep -E "^voxel|tv |GVoxel" gpurun_out/ab/vox_${L}_$rep.txt | tr '\n' ';' | sed 's/  */ /g' | cut -c1-900; echo
  done
done
