#!/bin/bash
# kernel-trace of a short bench run; prints per-kernel average durations
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-trace}
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/$TAG -o $TAG -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof/bench_$TAG.json 2> gpurun_out/prof/$TAG.err
python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/prof/$TAG/*kernel_trace.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%8.1f us avg  n=%5d  total %9.1f  %s" % (sum(v) / len(v), len(v), sum(v), k))
PY
