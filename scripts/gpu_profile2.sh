#!/bin/bash
# rocprofv3 kernel-trace summary + PMC passes (TCC: FETCH_SIZE, WRITE_SIZE in separate passes; SQ) of the bench, voxel kernels
# included, + the HIP-only synthetic training run.  Results under gpurun_out/{prof,pmc}; scripts/make_profile_summary.py TAG
# turns them into profiles/TAG_*.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r02}
mkdir -p gpurun_out/prof gpurun_out/pmc
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -2 gpurun_out/bench_$TAG.err; cut -c1-1500 gpurun_out/bench_$TAG.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/$TAG -o $TAG -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/prof/bench_prof_$TAG.json 2> gpurun_out/prof/rocprof_$TAG.err
tail -2 gpurun_out/prof/rocprof_$TAG.err
F=$(find gpurun_out/prof/$TAG -name "*kernel_stats.csv" | head -1); cp "$F" gpurun_out/prof/$TAG/${TAG}_kernel_stats.csv 2>/dev/null; head -12 "$F"
find gpurun_out/prof/$TAG -name "*kernel_trace.csv" -size +20M -delete
CMD="python bench.py --steps 10 --warmup 3 --no-cpu-baseline" bash scripts/gpu_pmc.sh $TAG 2>&1 | tail -30
timeout 900 python scripts/train_synthetic.py --iterations 3000 2>&1 | tail -14
