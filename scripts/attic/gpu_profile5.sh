#!/bin/bash
# Round-5 profile set (same as round 4's + the new bench sections come with the bench lines): bench lines (driver-style, default, trained clouds, configs B / C / E, one-rank collective path), rocprofv3
# kernel stats and TCC / SQ counter passes of the HEADLINE STEP ONLY (bench.py --headline-only) and of the 256^3 voxel query alone,
# the C harness with in-kernel stamps, the HIP-only trainer.  scripts/make_profile_summary5.py TAG turns them into profiles/TAG_*.
#   gpurun -- bash scripts/gpu_profile5.sh r05f
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05f}
mkdir -p gpurun_out/prof gpurun_out/pmc
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_driver.json 2> gpurun_out/bench_${TAG}_driver.err; tail -2 gpurun_out/bench_${TAG}_driver.err; cut -c1-200 gpurun_out/bench_${TAG}_driver.json
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -2 gpurun_out/bench_$TAG.err; cut -c1-200 gpurun_out/bench_$TAG.json
for C in small large; do
  timeout 600 python bench.py --cloud $C > gpurun_out/bench_${TAG}_trained_$C.json 2> gpurun_out/bench_${TAG}_trained_$C.err; cut -c1-200 gpurun_out/bench_${TAG}_trained_$C.json
done
HL="python bench.py --headline-only --steps 30 --warmup 5 --repeats 3"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/$TAG -o $TAG -- $HL > gpurun_out/prof/bench_prof_$TAG.json 2> gpurun_out/prof/rocprof_$TAG.err
tail -1 gpurun_out/prof/rocprof_$TAG.err
F=$(find gpurun_out/prof/$TAG -name "*kernel_stats.csv" | head -1); cp "$F" gpurun_out/prof/$TAG/${TAG}_kernel_stats.csv 2>/dev/null; head -10 "$F" | cut -c1-160
find gpurun_out/prof/$TAG -name "*kernel_trace.csv" -size +20M -delete
CMD="$HL" bash scripts/gpu_pmc.sh $TAG 2>&1 | tail -14
CMD="$HL" bash scripts/gpu_pmc2.sh $TAG 2>&1 | tail -4
# the voxelizer alone (256^3 query): kernel stats + counters
VQ="python scripts/voxel_query_only.py 12"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${TAG}_vox -o vox -- $VQ > /dev/null 2> gpurun_out/prof/rocprof_${TAG}_vox.err
F=$(find gpurun_out/prof/${TAG}_vox -name "*kernel_stats.csv" | head -1); cp "$F" gpurun_out/prof/${TAG}_vox_kernel_stats.csv 2>/dev/null; head -8 "$F" | cut -c1-140
find gpurun_out/prof/${TAG}_vox -name "*kernel_trace.csv" -size +20M -delete
CMD="$VQ" bash scripts/gpu_pmc.sh ${TAG}_vox 2>&1 | tail -10
# the other BASELINE configurations (B: 50k / 512^2, C: 300k / 560^2, E: 1M / 1024^2 / 360 views), raster only
timeout 300 python bench.py --gaussians 50000 --no-voxel --no-streams --no-batched > gpurun_out/bench_${TAG}_B.json 2>/dev/null; cut -c1-160 gpurun_out/bench_${TAG}_B.json
timeout 300 python bench.py --detector 560 --no-voxel --no-streams --no-batched > gpurun_out/bench_${TAG}_C.json 2>/dev/null; cut -c1-160 gpurun_out/bench_${TAG}_C.json
timeout 600 python bench.py --gaussians 1000000 --detector 1024 --views 360 --steps 300 --warmup 30 --no-voxel --no-streams --no-batched > gpurun_out/bench_${TAG}_E.json 2>/dev/null; cut -c1-160 gpurun_out/bench_${TAG}_E.json
# one rank through the collective path (RCCL initialised, the [11 P] block all-reduced with nobody to talk to)
R2_BENCH_FORCE_COMM=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 200 --warmup 20 --no-batched --no-streams --no-forward-only --no-cpu-baseline --no-densify-pattern > gpurun_out/bench_${TAG}_forcecomm.json 2> gpurun_out/bench_${TAG}_forcecomm.err; cut -c1-160 gpurun_out/bench_${TAG}_forcecomm.json
# C host for the ABI: A/B of the two binning chains, then the stamped build's timeline
for E in 0 1; do R2_TILE_FIRST=$E timeout 300 scripts/cbench 300 r2_gaussian_amd/libr2hip.so > gpurun_out/cbench_${TAG}_tf$E.txt 2>&1; grep -E "BEST|STREAMS|BATCH V=4:|^voxel|raster\.|tv " gpurun_out/cbench_${TAG}_tf$E.txt; done
[ -f r2_gaussian_amd/libr2hip_ts.so ] && timeout 100 scripts/cbench 100 r2_gaussian_amd/libr2hip_ts.so single 2>&1 | grep -E "BEST|TS " > gpurun_out/timeline_${TAG}.txt
# HIP-only training run (fused losses / densify), PSNR + it/s
timeout 900 python scripts/train_synthetic.py --iterations 3000 --fused-losses 2>&1 | tail -2
