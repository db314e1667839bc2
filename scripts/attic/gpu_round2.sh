#!/bin/bash
# one gpurun call: GPU parity tests (all of them, no -x, parity margins -> gpurun_out/parity_report.json) + smoke +
# bench (driver-style short run and the default run) + the C-ABI timeline harness
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r02a}
mkdir -p gpurun_out
nproc; free -g | head -2
timeout 1500 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -40 | tee gpurun_out/pytest_$TAG.log
cp gpurun_out/parity_report.json gpurun_out/parity_report_$TAG.json 2>/dev/null
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/smoke_$TAG.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_driver.json 2> gpurun_out/bench_${TAG}_driver.err; tail -3 gpurun_out/bench_${TAG}_driver.err; cat gpurun_out/bench_${TAG}_driver.json
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -3 gpurun_out/bench_$TAG.err; cat gpurun_out/bench_$TAG.json
if [ -x scripts/cbench ]; then timeout 300 scripts/cbench 300 > gpurun_out/cbench_$TAG.txt 2>&1; head -40 gpurun_out/cbench_$TAG.txt; fi
