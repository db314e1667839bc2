#!/bin/bash
# Round-3 profile set: bench lines (default, driver-style, configs B / C / E), rocprofv3 kernel stats and TCC / SQ counter passes
# of the HEADLINE STEP ONLY (bench.py --headline-only: every kernel row is the single-view forward + backward), the voxelizer
# alone, the C harness.  Results under gpurun_out/{prof,pmc}; scripts/make_profile_summary3.py TAG turns them into profiles/TAG_*.
#   gpurun -- bash scripts/gpu_profile3.sh r03
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r03}
mkdir -p gpurun_out/prof gpurun_out/pmc
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_driver.json 2> gpurun_out/bench_${TAG}_driver.err; tail -2 gpurun_out/bench_${TAG}_driver.err; cut -c1-300 gpurun_out/bench_${TAG}_driver.json
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -2 gpurun_out/bench_$TAG.err; cut -c1-300 gpurun_out/bench_$TAG.json
HL="python bench.py --headline-only --steps 30 --warmup 5 --repeats 3"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/$TAG -o $TAG -- $HL > gpurun_out/prof/bench_prof_$TAG.json 2> gpurun_out/prof/rocprof_$TAG.err
tail -1 gpurun_out/prof/rocprof_$TAG.err
F=$(find gpurun_out/prof/$TAG -name "*kernel_stats.csv" | head -1); cp "$F" gpurun_out/prof/$TAG/${TAG}_kernel_stats.csv 2>/dev/null; head -8 "$F" | cut -c1-160
find gpurun_out/prof/$TAG -name "*kernel_trace.csv" -size +20M -delete
CMD="$HL" bash scripts/gpu_pmc.sh $TAG 2>&1 | tail -16
CMD="$HL" bash scripts/gpu_pmc2.sh $TAG 2>&1 | tail -4
# the other BASELINE configurations (B: 50k / 512^2, C: 300k / 560^2, E: 1M / 1024^2 / 360 views), raster only
timeout 300 python bench.py --gaussians 50000 --no-voxel --no-streams --no-batched > gpurun_out/bench_${TAG}_B.json 2>/dev/null; cut -c1-200 gpurun_out/bench_${TAG}_B.json
timeout 300 python bench.py --detector 560 --no-voxel --no-streams --no-batched > gpurun_out/bench_${TAG}_C.json 2>/dev/null; cut -c1-200 gpurun_out/bench_${TAG}_C.json
timeout 600 python bench.py --gaussians 1000000 --detector 1024 --views 360 --steps 300 --warmup 30 --no-voxel --no-streams --no-batched > gpurun_out/bench_${TAG}_E.json 2>/dev/null; cut -c1-200 gpurun_out/bench_${TAG}_E.json
# the voxelizer alone (256^3 query): kernel stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${TAG}_vox -o vox -- python scripts/voxel_query_only.py 12 > /dev/null 2> gpurun_out/prof/rocprof_${TAG}_vox.err
F=$(find gpurun_out/prof/${TAG}_vox -name "*kernel_stats.csv" | head -1); cp "$F" gpurun_out/prof/${TAG}_vox_kernel_stats.csv 2>/dev/null; head -8 "$F" | cut -c1-140
find gpurun_out/prof/${TAG}_vox -name "*kernel_trace.csv" -size +20M -delete
# C host for the ABI
timeout 300 scripts/cbench 300 > gpurun_out/cbench_$TAG.txt 2>&1; grep -E "BEST|STREAMS|BATCH V=4:|^voxel|raster\.|tv " gpurun_out/cbench_$TAG.txt
# HIP-only training run (fused losses / densify), PSNR + it/s
timeout 900 python scripts/train_synthetic.py --iterations 3000 --fused-losses 2>&1 | tail -3
