#!/bin/bash
# the whole GPU suite twice: in collection order and with the test files in reverse order (per-thread prediction / hint history
# left by one test must not decide another's outcome), + smoke
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05f}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --durations=5 2>&1 | tail -14 | tee gpurun_out/pytest_$TAG.log
cp gpurun_out/parity_report.json gpurun_out/parity_report_$TAG.json 2>/dev/null
timeout 1500 python -m pytest $(ls tests/test_*_gpu.py | sort -r) -q -m gpu -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/pytest_${TAG}_reversed.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/smoke_$TAG.log
