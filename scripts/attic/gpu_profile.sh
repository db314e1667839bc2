#!/bin/bash
# rocprofv3 kernel-trace summary of the headline bench (run through gpurun; results land in gpurun_out/)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
TAG=${1:-r01}
python bench.py --steps 100 --warmup 20 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; cat gpurun_out/bench_$TAG.json
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/$TAG -o $TAG -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/prof/bench_prof_$TAG.json 2> gpurun_out/prof/rocprof_$TAG.err
tail -2 gpurun_out/prof/rocprof_$TAG.err
find gpurun_out/prof/$TAG -name "*stats*" | head
F=$(find gpurun_out/prof/$TAG -name "*kernel_stats.csv" | head -1)
head -30 "$F"
