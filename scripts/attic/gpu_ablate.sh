#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
for i in 0 1 3; do
  L=$GRAFT_REPO_ROOT/r2_gaussian_amd/libr2hip_abl$i.so; [ $i = 0 ] && L=$GRAFT_REPO_ROOT/r2_gaussian_amd/libr2hip.so
  echo "== ABL $i"; R2HIP_LIB=$L python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-voxel | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k:v['us'] for k,v in d['kernels'].items()})"
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/prof/pmc1 -o pmc1 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-voxel > /dev/null 2> gpurun_out/prof/pmc1.err
tail -2 gpurun_out/prof/pmc1.err; ls gpurun_out/prof/pmc1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof/pmc1/*counter_collection.csv")
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:60]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
    for k in agg:
        if "render" in k or "sort" in k.lower() or "onesweep" in k.lower():
            print(k, n[k], {c: round(v / max(n[k],1)) for c, v in agg[k].items()})
PY
