#!/bin/bash
# run-to-run noise of the headline bench under a few host settings
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-voxel 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['host_wait_us_per_step'])"; }
nproc; lscpu | grep -E "Model name|Socket|NUMA node\(s\)|Thread" | head -5
for i in 1 2 3; do run base; done
for i in 1 2 3; do R2_BENCH_NOGC=1 run nogc; done
for i in 1 2 3; do taskset -c 0-15 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-voxel 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pin0-15', d['value'], d['ms_per_step'], d['host_wait_us_per_step'])"; done
for i in 1 2 3; do GPU_MAX_HW_QUEUES=2 run hwq2; done
