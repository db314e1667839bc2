#!/bin/bash
# round 4: one gpurun call = GPU suite (incl. the trained-cloud parity cases) + smoke + bench on the synthetic cloud and on the
# two trained clouds (tests/trained_cloud.py trains them on this box when gpurun_out/clouds did not travel)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r04a}
mkdir -p gpurun_out
nproc; free -g | head -2
if [ "$2" != "notests" ]; then
timeout 2400 python -m pytest tests -q -m gpu --durations=12 2>&1 | tail -45 | tee gpurun_out/pytest_$TAG.log
cp gpurun_out/parity_report.json gpurun_out/parity_report_$TAG.json 2>/dev/null
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/smoke_$TAG.log
fi
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_driver.json 2> gpurun_out/bench_${TAG}_driver.err; tail -3 gpurun_out/bench_${TAG}_driver.err; cat gpurun_out/bench_${TAG}_driver.json
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -3 gpurun_out/bench_$TAG.err; cat gpurun_out/bench_$TAG.json
for C in small large; do
  timeout 900 python bench.py --cloud $C > gpurun_out/bench_${TAG}_trained_$C.json 2> gpurun_out/bench_${TAG}_trained_$C.err
  tail -3 gpurun_out/bench_${TAG}_trained_$C.err; cat gpurun_out/bench_${TAG}_trained_$C.json
done
if [ -x scripts/cbench ]; then timeout 300 scripts/cbench 300 > gpurun_out/cbench_$TAG.txt 2>&1; head -40 gpurun_out/cbench_$TAG.txt; fi
