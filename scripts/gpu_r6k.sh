#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_raster_gpu.py tests/test_tilefirst_gpu.py tests/test_batch_gpu.py tests/test_autograd_gpu.py tests/test_variants_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee gpurun_out/pytest_r6k.log
TAG=${TAG:-r6k} LIBS="${LIBS:-libr2hip_prev.so libr2hip.so}" bash scripts/gpu_ab7.sh
