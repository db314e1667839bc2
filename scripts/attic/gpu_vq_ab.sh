#!/bin/bash
# kernel stats of the 256^3 voxel query alone for several library builds, same box:
#   gpurun -- bash scripts/gpu_vq_ab.sh "libr2hip_prev.so libr2hip.so"
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/vq
for L in ${1:-libr2hip.so}; do
  echo "=== $L"
  rm -rf /tmp/vq_$L
  R2HIP_LIB=$GRAFT_REPO_ROOT/r2_gaussian_amd/$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vq_$L -o vq -- python scripts/voxel_query_only.py 12 > /dev/null 2> gpurun_out/vq/err_$L.txt; tail -2 gpurun_out/vq/err_$L.txt
  F=$(find /tmp/vq_$L -name "*kernel_stats.csv" | head -1)
  cp $F gpurun_out/vq/${L}_kernel_stats.csv
  python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
for r in rows:
    n = int(r["Calls"]); avg = float(r["AverageNs"]) / 1e3
    per = float(r["TotalDurationNs"]) / 1e3 / 12.0
    tot += per
    if per > 3.0:
        print("%-60s calls %3d avg %8.1f us  per query %8.1f us" % (r["Name"][:60], n, avg, per))
print("sum per query %.1f us" % tot)
PY
done
