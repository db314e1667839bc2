#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --durations=5 -x 2>&1 | tail -14 | tee gpurun_out/pytest_r6j.log
TAG=r6j LIBS="libr2hip_base.so libr2hip.so" bash scripts/gpu_ab7.sh
