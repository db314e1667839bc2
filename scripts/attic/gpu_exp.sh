#!/bin/bash
# quick experiment call: parity tests (subset via $TESTS) + the C-ABI harness timing
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-exp}
TESTS=${TESTS:-"tests/test_raster_gpu.py tests/test_golden_gpu.py tests/test_reported_configs_gpu.py tests/test_sort_gpu.py"}
if [ "$TESTS" != "none" ]; then timeout 900 python -m pytest $TESTS -q -m gpu -x 2>&1 | tail -6; fi
for rep in 1 2; do timeout 300 scripts/cbench ${STEPS:-300} > gpurun_out/cbench_$TAG.txt 2>&1; grep -E "BEST|raster\.|voxel 256|voxel 32|BATCH" gpurun_out/cbench_$TAG.txt | grep -v "V=" | head -14; grep -E "BATCH" gpurun_out/cbench_$TAG.txt; done
grep "V=4 " gpurun_out/cbench_$TAG.txt
