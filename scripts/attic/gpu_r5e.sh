#!/bin/bash
# round 5 checkpoint: the whole GPU suite, the bench line with its new sections, A/B against the round-4 library, voxel query counters
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05a}
mkdir -p gpurun_out/ab gpurun_out/ts
timeout 1500 python -m pytest tests -q -m gpu --durations=6 2>&1 | tail -30 | tee gpurun_out/pytest_$TAG.log
cp gpurun_out/parity_report.json gpurun_out/parity_report_$TAG.json 2>/dev/null
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -3 gpurun_out/bench_$TAG.err; cut -c1-400 gpurun_out/bench_$TAG.json
for rep in 1 2; do
  for L in libr2hip_base.so libr2hip.so; do
    timeout 200 scripts/cbench 300 r2_gaussian_amd/$L single,stages,voxel > gpurun_out/ab/${L}_$rep.txt 2>&1
    echo "== $L (rep $rep): $(grep -E 'BEST|raster\.|^voxel|GVoxel' gpurun_out/ab/${L}_$rep.txt | tr '\n' ';' | sed 's/  */ /g' | cut -c1-700)"
  done
done
timeout 200 scripts/cbench 100 r2_gaussian_amd/libr2hip_ts.so single > gpurun_out/ts/libr2hip_ts.so.txt 2>&1
echo "== stamps"; grep -E "BEST|TS (geom|tilefirst)" gpurun_out/ts/libr2hip_ts.so.txt
CMD="python scripts/voxel_query_only.py 12" COUNTERS="WRITE_SIZE" bash scripts/gpu_pmc.sh ${TAG}_vox 2>&1 | grep -E "voxel_render|FETCH|no counter" | head
