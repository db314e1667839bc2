#!/bin/bash
# kernel timeline of the C harness: per-kernel start/end -> durations and idle gaps of one steady-state step
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o tl -- scripts/cbench ${1:-60} r2_gaussian_amd/libr2hip.so ${2:-single} > gpurun_out/tl/cbench.txt 2>&1
python3 - <<'PY'
import csv, glob, re, collections
f = glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r'(r2::(?:\(anonymous namespace\)::)?[a-zA-Z_0-9]+)', n)
    return m.group(1).replace('(anonymous namespace)::', '').replace('r2::', '') if m else n.split('(')[0][:40]
names = [short(r["Kernel_Name"]) for r in rows]
# steps start at raster_preprocess_kernel; take steady-state steps that are followed by a geom backward
starts = [i for i, n in enumerate(names) if n == "raster_preprocess_kernel"]
steps = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    if any(short(r["Kernel_Name"]) == "raster_geom_backward_kernel" for r in seg):
        steps.append((a, b))
steps = steps[30:80]
agg = collections.OrderedDict()
tot = 0.0
for a, b in steps:
    prev_end = None
    for i in range(a, b):
        r = rows[i]; n = names[i] + "#%d" % sum(1 for j in range(a, i) if names[j] == names[i])
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        d = agg.setdefault(n, [0.0, 0.0, 0])
        d[0] += (e - s) / 1e3
        d[1] += ((s - prev_end) / 1e3) if prev_end is not None else 0.0
        d[2] += 1
        prev_end = e
    tot += (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
print("steps analysed:", len(steps), " mean step (start to start): %.1f us" % (tot / max(len(steps), 1)))
print("%-34s %8s %8s" % ("kernel (in launch order)", "dur us", "gap us"))
sd = sg = 0
for n, (d, g, c) in agg.items():
    print("%-34s %8.1f %8.1f" % (n, d / c, g / c)); sd += d / c; sg += g / c
print("%-34s %8.1f %8.1f" % ("sum", sd, sg))
PY
find gpurun_out/tl -name "*.csv" -size +4M -delete
