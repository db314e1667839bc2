#!/bin/bash
# second SQ counter set (LDS / memory instruction mix) for a command; usage: CMD="..." bash scripts/gpu_pmc2.sh TAG
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-x}
mkdir -p gpurun_out/pmc
CMD=${CMD:-"python bench.py --steps 3 --warmup 2 --no-cpu-baseline"}
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc/${TAG}_SQ2 -o SQ2 -- $CMD > /dev/null 2> gpurun_out/pmc/${TAG}_SQ2.err
tail -2 gpurun_out/pmc/${TAG}_SQ2.err
python - <<PY
import csv, glob, collections, re
def short(n):
    m = re.search(r'(r2::(?:\(anonymous namespace\)::)?[a-zA-Z_0-9]+(?:<[^>(]*>)?)', n)
    return m.group(1).replace('(anonymous namespace)::', '') if m else n.split('(')[0][:60]
fs = glob.glob("gpurun_out/pmc/${TAG}_SQ2/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(fs[0])):
    k = short(r["Kernel_Name"]); agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in agg:
    if "render" in k:
        print(k, {c: round(v / max(n[k][c], 1)) for c, v in agg[k].items()})
PY
find gpurun_out/pmc -name "*.csv" -size +8M -delete
