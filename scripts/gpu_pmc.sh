#!/bin/bash
# HBM traffic of every kernel from the TCC counters, one counter per pass (FETCH_SIZE costs 3 of the 4 TCC slots,
# WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots"), plus an SQ pass for the render kernels.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r01}
mkdir -p gpurun_out/pmc
CMD=${CMD:-"python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-voxel"}
for C in ${COUNTERS:-FETCH_SIZE WRITE_SIZE}; do
  timeout 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc/${TAG}_$C -o $C -- $CMD > /dev/null 2> gpurun_out/pmc/${TAG}_$C.err
  tail -1 gpurun_out/pmc/${TAG}_$C.err
done
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc/${TAG}_SQ -o SQ -- $CMD > /dev/null 2> gpurun_out/pmc/${TAG}_SQ.err
python - <<PY
import csv, glob, collections, json, re
def short(n):
    m = re.search(r'(r2::(?:\(anonymous namespace\)::)?[a-zA-Z_0-9]+(?:<[^>(]*>)?)', n)
    return m.group(1).replace('(anonymous namespace)::', '') if m else n.split('(')[0][:60]
res = collections.defaultdict(dict)
for C in ("FETCH_SIZE", "WRITE_SIZE", "SQ"):
    fs = glob.glob("gpurun_out/pmc/${TAG}_%s/**/*counter_collection.csv" % C, recursive=True)
    if not fs:
        print("no counter csv for", C); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(fs[0])):
        k = short(r["Kernel_Name"]); agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    for k in agg:
        for c, v in agg[k].items():
            res[k][c] = v / max(n[k][c], 1)
            res[k]["launches_" + c] = n[k][c]
import sys
sys.path.insert(0, ".")
from r2_gaussian_amd import build as _b
res["_meta"] = {"source_sha": _b.source_hash(), "cmd": """$CMD""", "tag": "${TAG}"}   # which kernels these counters belong to
json.dump(res, open("gpurun_out/pmc/${TAG}_pmc_per_launch.json", "w"), indent=1, sort_keys=True)
res.pop("_meta")
for k, d in sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
    if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
        print("%-46s FETCH_SIZE %10.1f KB  WRITE_SIZE %10.1f KB  waves %s valu %s" % (k, d.get("FETCH_SIZE", -1), d.get("WRITE_SIZE", -1), d.get("SQ_WAVES"), d.get("SQ_INSTS_VALU")))
PY
find gpurun_out/pmc -name "*.csv" -size +8M -delete
