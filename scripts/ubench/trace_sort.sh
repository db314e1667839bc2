cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof/sortb -o sortb -- ./scripts/ubench/sort_bench > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof/sortb/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# print the last sort of the first case: find sequences
prev_end = None
out = []
for r in rows[60:100]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    out.append("%-40s dur %7.1f us  gap %6.1f us  grid %s" % (r["Kernel_Name"][:40], (e - s) / 1e3, gap, r["Grid_Size_X"]))
    prev_end = e
print("\n".join(out))
PY
