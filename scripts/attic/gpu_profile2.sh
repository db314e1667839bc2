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
timeout 900 python scripts/train_synthetic.py --iterations 3000 --fused-losses 2>&1 | tail -6
# the other BASELINE configurations (B: 50k / 512^2, E: 1M / 1024^2 / 360 views), same bench, raster only
timeout 300 python bench.py --gaussians 50000 --no-voxel > gpurun_out/bench_${TAG}_B.json 2>/dev/null; cut -c1-400 gpurun_out/bench_${TAG}_B.json
timeout 600 python bench.py --gaussians 1000000 --detector 1024 --views 360 --steps 300 --warmup 30 --no-voxel > gpurun_out/bench_${TAG}_E.json 2>/dev/null; cut -c1-400 gpurun_out/bench_${TAG}_E.json
# C host for the ABI: this build and (if present) the round-1 library on the same box
for L in libr2hip_r01.so libr2hip.so; do [ -f r2_gaussian_amd/$L ] && { echo "== $L"; timeout 200 scripts/cbench 300 r2_gaussian_amd/$L > gpurun_out/cbench_${TAG}_$L.txt 2>&1; grep -E "BEST|raster\.|^voxel|BATCH" gpurun_out/cbench_${TAG}_$L.txt | grep -v " V="; }; done
# device-side timeline (experiment build with s_memrealtime stamps), C host and Python host
[ -f r2_gaussian_amd/libr2hip_ts.so ] && { timeout 100 scripts/cbench 100 r2_gaussian_amd/libr2hip_ts.so 2>&1 | grep -E "BEST|TS " > gpurun_out/timeline_${TAG}_c.txt; R2HIP_LIB=$PWD/r2_gaussian_amd/libr2hip_ts.so timeout 200 python scripts/ts_python.py 2>&1 | grep "TS " > gpurun_out/timeline_${TAG}_py.txt; wc -l gpurun_out/timeline_${TAG}_*.txt; }
