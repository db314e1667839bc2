#!/bin/bash
# one gpurun call: GPU parity tests + smoke + headline bench + rocprofv3 kernel-trace summary
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r01}
mkdir -p gpurun_out/prof
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee gpurun_out/pytest_$TAG.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/smoke_$TAG.log
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -3 gpurun_out/bench_$TAG.err; cat gpurun_out/bench_$TAG.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/$TAG -o $TAG -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/prof/bench_prof_$TAG.json 2> gpurun_out/prof/rocprof_$TAG.err
tail -2 gpurun_out/prof/rocprof_$TAG.err
F=$(find gpurun_out/prof/$TAG -name "*kernel_stats.csv" | head -1)
head -40 "$F"
# keep only the summaries (traces are big)
find gpurun_out/prof/$TAG -name "*kernel_trace.csv" -size +20M -delete
