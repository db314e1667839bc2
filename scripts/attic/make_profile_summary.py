"""profiles/<tag>_summary.md from the rocprofv3 kernel stats + per-launch PMC json that scripts/gpu_round.sh and
scripts/gpu_pmc.sh leave under gpurun_out/:   python scripts/make_profile_summary.py r01r "title" """
import csv, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
title = sys.argv[2] if len(sys.argv) > 2 else tag
P = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(ROOT, "gpurun_out/prof/%s/%s_kernel_stats.csv" % (tag, tag)), os.path.join(P, "%s_kernel_stats.csv" % tag))
shutil.copy(os.path.join(ROOT, "gpurun_out/pmc/%s_pmc_per_launch.json" % tag), os.path.join(P, "%s_pmc.json" % tag))
shutil.copy(os.path.join(ROOT, "gpurun_out/pmc/%s_pmc_per_launch.json" % tag), os.path.join(P, "pmc_latest.json"))
b = os.path.join(ROOT, "gpurun_out/bench_%s.json" % tag)
if os.path.exists(b):
    shutil.copy(b, os.path.join(P, "%s_bench.json" % tag))


def short(n):
    m = re.search(r'(r2::(?:\(anonymous namespace\)::)?[a-zA-Z_0-9]+(?:<[^>(]*>)?)', n)
    return m.group(1).replace('(anonymous namespace)::', '') if m else n.split('(')[0][:70]


rows = list(csv.DictReader(open(os.path.join(P, "%s_kernel_stats.csv" % tag))))
pmc = json.load(open(os.path.join(P, "%s_pmc.json" % tag)))
with open(os.path.join(P, "%s_summary.md" % tag), "w") as f:
    f.write("# %s\n\n" % title)
    f.write("`rocprofv3 --kernel-trace --stats -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline` on MI355X "
            "(full CSV: %s_kernel_stats.csv).  PMC columns: separate rocprofv3 passes, one TCC counter per pass "
            "(scripts/gpu_pmc.sh, `bench.py --no-voxel`), averaged per launch (%s_pmc.json = pmc_latest.json, which bench.py "
            "reads for `roofline.traffic`); FETCH_SIZE / WRITE_SIZE are KB in the json, MB here.  Bench line of the same "
            "build: %s_bench.json.\n\n" % (tag, tag, tag))
    f.write("| kernel | calls | avg us | % | FETCH MB | WRITE MB | VALU inst (M) |\n|---|---|---|---|---|---|---|\n")
    for r in rows[:36]:
        k = short(r['Name']); p = pmc.get(k, {})
        f.write("| `%s` | %s | %.1f | %s | %s | %s | %s |\n" % (
            k, r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage'],
            ("%.1f" % (p['FETCH_SIZE'] / 1024)) if 'FETCH_SIZE' in p else "",
            ("%.1f" % (p['WRITE_SIZE'] / 1024)) if 'WRITE_SIZE' in p else "",
            ("%.1f" % (p['SQ_INSTS_VALU'] / 1e6)) if 'SQ_INSTS_VALU' in p else ""))
    f.write("\nNotes: FETCH_SIZE on gfx950 under-reports wide (16 B/lane) streaming reads by 2x (MI355X_MICROARCH.md, HBM); the "
            "render kernels gather 16-byte records, so their true fetch lies between the reported value and twice it.  The calls "
            "column mixes the bench's sections: `<false>` kernels are the single-view step (`value`), `<true>` ones the batched "
            "calls (V = 4), voxel kernels the 256^3 query and the 32^3 TV patch; average durations of kernels used by several "
            "sections are averages over all of them.  Kernels named bucket_* / minmax / scan_* belong to the un-hinted "
            "depth order (first call for a given P, and the separate `_C` calls bench.py makes to count num_rendered).\n")
print(open(os.path.join(P, "%s_summary.md" % tag)).read()[:1800])
