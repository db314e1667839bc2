#!/bin/bash
# kernel stats + TCC / SQ counters of the 256^3 voxel query alone (the voxel part of gpu_profile4.sh): gpurun -- bash scripts/gpu_vq_profile.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r04m}
mkdir -p gpurun_out/prof gpurun_out/pmc
VQ="python scripts/voxel_query_only.py 12"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${TAG}_vox -o vox -- $VQ > /dev/null 2> gpurun_out/prof/rocprof_${TAG}_vox.err
F=$(find gpurun_out/prof/${TAG}_vox -name "*kernel_stats.csv" | head -1); cp "$F" gpurun_out/prof/${TAG}_vox_kernel_stats.csv 2>/dev/null; head -8 "$F" | cut -c1-140
find gpurun_out/prof/${TAG}_vox -name "*kernel_trace.csv" -size +20M -delete
CMD="$VQ" bash scripts/gpu_pmc.sh ${TAG}_vox 2>&1 | tail -6
