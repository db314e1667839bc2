#!/bin/bash
# round 6: the one-wave forward kernel -- rasterizer suites, then A/B through the C host (base = the tree before it)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/ab
timeout 1500 python -m pytest tests/test_raster_gpu.py tests/test_tilefirst_gpu.py tests/test_batch_gpu.py tests/test_variants_gpu.py tests/test_autograd_gpu.py tests/test_golden_gpu.py tests/test_threads_gpu.py tests/test_multistream_gpu.py -q -m gpu -x --durations=5 2>&1 | tail -25 | tee gpurun_out/pytest_r6c.log
for rep in 1 2; do
  for L in ${LIBS:-libr2hip_base.so libr2hip.so R2_FWD_WAVE=0:libr2hip.so}; do
    E=""; F=$L; case $L in *:*) E=${L%%:*}; F=${L#*:};; esac
    env $E timeout 200 scripts/cbench ${STEPS:-300} r2_gaussian_amd/$F single,stages > gpurun_out/ab/r6c_${L}_$rep.txt 2>&1
    echo "== $L (rep $rep): $(grep -E 'BEST|raster\.' gpurun_out/ab/r6c_${L}_$rep.txt | tr '\n' ';' | sed 's/  */ /g')"
  done
done | tee gpurun_out/ab/r6c_summary.txt
