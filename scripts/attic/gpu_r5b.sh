#!/bin/bash
# in-kernel stamps (-DR2_EXP_TS) of the round-4 chain (libr2hip_ts_base.so) and of the two-kernel chain (libr2hip_ts.so)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/ts
for L in libr2hip_ts_base.so R2_TF_SLABS=1:libr2hip_ts.so; do
  E=""; F=$L; case $L in *:*) E=${L%%:*}; F=${L#*:};; esac
  env $E timeout 200 scripts/cbench 100 r2_gaussian_amd/$F single > gpurun_out/ts/$F.txt 2>&1
  echo "== $L"; grep -E "BEST|TS (geom|tilefirst|render)" gpurun_out/ts/$F.txt
done
