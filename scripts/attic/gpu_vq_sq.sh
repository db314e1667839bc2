#!/bin/bash
# SQ counters of the 256^3 voxel query's render kernel for several library builds: bash scripts/gpu_vq_sq.sh "libA.so libB.so"
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/vq
for L in ${1:-libr2hip.so}; do
  rm -rf /tmp/sq_$L
  R2HIP_LIB=$GRAFT_REPO_ROOT/r2_gaussian_amd/$L timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d /tmp/sq_$L -o SQ -- python scripts/voxel_query_only.py 6 > /dev/null 2> gpurun_out/vq/sqerr_$L.txt
  python - "$L" <<'PY'
import csv, glob, collections, sys
L = sys.argv[1]
fs = glob.glob("/tmp/sq_%s/**/*counter_collection.csv" % L, recursive=True)
agg = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(fs[0])):
    if "voxel_render_forward_both" in r["Kernel_Name"] or "voxel_render_backward" in r["Kernel_Name"]:
        k = ("fwd" if "forward" in r["Kernel_Name"] else "bwd", r["Counter_Name"]); agg[k] += float(r["Counter_Value"]); n[k] += 1
print(L, {"%s.%s" % k: round(v / n[k] / 1e6, 2) for k, v in sorted(agg.items())})
PY
done
