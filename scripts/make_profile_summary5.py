"""profiles/<tag>_* from what scripts/gpu_profile6.sh (rounds 4-5: gpu_profile5.sh, now in scripts/attic) left under gpurun_out/ (rounds 4-5: rocprofv3 passes over
`bench.py --headline-only`, so every kernel row is the single-view forward + backward step, + the same for the 256^3 voxel
query alone, + the bench lines of the trained clouds and of the one-rank collective path):
    python scripts/make_profile_summary5.py r05f "title"
"""
import csv, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
title = sys.argv[2] if len(sys.argv) > 2 else tag
P = os.path.join(ROOT, "profiles")
G = os.path.join(ROOT, "gpurun_out")


def cp(src, dst):
    if os.path.exists(src):
        shutil.copy(src, dst)
        return True
    return False


cp(os.path.join(G, "prof/%s/%s_kernel_stats.csv" % (tag, tag)), os.path.join(P, "%s_kernel_stats.csv" % tag))
cp(os.path.join(G, "prof/%s_vox_kernel_stats.csv" % tag), os.path.join(P, "%s_voxel256_kernel_stats.csv" % tag))
cp(os.path.join(G, "pmc/%s_pmc_per_launch.json" % tag), os.path.join(P, "%s_pmc.json" % tag))
cp(os.path.join(G, "pmc/%s_pmc_per_launch.json" % tag), os.path.join(P, "pmc_latest.json"))
cp(os.path.join(G, "pmc/%s_vox_pmc_per_launch.json" % tag), os.path.join(P, "%s_voxel256_pmc.json" % tag))
for suffix in ("_tf0", "_tf1"):
    cp(os.path.join(G, "cbench_%s%s.txt" % (tag, suffix)), os.path.join(P, "%s_cbench%s.txt" % (tag, suffix)))
cp(os.path.join(G, "timeline_%s.txt" % tag), os.path.join(P, "%s_timeline_stamps.txt" % tag))
for suffix in ("", "_driver", "_B", "_C", "_E", "_trained_small", "_trained_large", "_forcecomm"):
    src, dst = os.path.join(G, "bench_%s%s.json" % (tag, suffix)), os.path.join(P, "%s_bench%s.json" % (tag, suffix))
    if os.path.exists(src):   # only the JSON line (RCCL prints a version banner to stdout in front of it)
        lines = [l for l in open(src).read().splitlines() if l.startswith('{"metric')]
        if lines:
            open(dst, "w").write(lines[-1] + "\n")
cp(os.path.join(G, "cbench_%s.txt" % tag), os.path.join(P, "%s_cbench.txt" % tag))
for f in ("train_synthetic_fused.json",):
    cp(os.path.join(G, f), os.path.join(P, "%s_%s" % (tag, f)))


def short(n):
    m = re.search(r'(r2::(?:\(anonymous namespace\)::)?[a-zA-Z_0-9]+(?:<[^>(]*>)?)', n)
    return m.group(1).replace('(anonymous namespace)::', '') if m else n.split('(')[0][:70]


if os.path.exists(os.path.join(P, "%s_kernel_stats.csv" % tag)):   # (a voxel-only set has no headline pass: scripts/gpu_vq_profile.sh)
    rows = list(csv.DictReader(open(os.path.join(P, "%s_kernel_stats.csv" % tag))))
    pmc = json.load(open(os.path.join(P, "%s_pmc.json" % tag)))
    with open(os.path.join(P, "%s_summary.md" % tag), "w") as f:
        f.write("# %s\n\n" % title)
        f.write("`rocprofv3 --kernel-trace --stats -- python bench.py --headline-only --steps 30 --warmup 5 --repeats 3` on MI355X: the "
                "single-view forward + backward step and nothing else (no batched / concurrent / voxelizer / CPU-baseline sections), so "
                "every row below IS the headline step (full CSV: %s_kernel_stats.csv).  PMC columns: separate rocprofv3 passes over the "
                "same command, one TCC counter per pass (FETCH_SIZE, WRITE_SIZE; SQ_* in a third), averaged per launch (%s_pmc.json = "
                "pmc_latest.json, which bench.py reads for `roofline.traffic` when its source hash matches); KB in the json, MB here.  "
                "Bench lines of the same build: %s_bench.json (default run), %s_bench_driver.json (--steps 20 --warmup 5), "
                "%s_bench_{B,C,E}.json (BASELINE configs B 50k/512^2, C 300k/560^2, E 1M/1024^2/360 views).\n\n" % (tag, tag, tag, tag, tag))
        f.write("| kernel | calls | avg us | % | FETCH MB | WRITE MB | VALU inst (M) | VALU floor frac |\n|---|---|---|---|---|---|---|---|\n")
        for r in rows[:30]:
            k = short(r['Name']); p = pmc.get(k, {})
            us = float(r['AverageNs']) / 1e3
            # fraction of the VALU floor: SQ_INSTS_VALU x 2.8 cycles / (1024 SIMDs x kernel cycles at the nominal 2.4 GHz)
            issue = (p['SQ_INSTS_VALU'] * 2.8 / (1024.0 * us * 2400.0)) if ('SQ_INSTS_VALU' in p and us > 0) else None
            f.write("| `%s` | %s | %.1f | %s | %s | %s | %s | %s |\n" % (
                k, r['Calls'], us, r['Percentage'],
                ("%.1f" % (p['FETCH_SIZE'] / 1024)) if 'FETCH_SIZE' in p else "",
                ("%.1f" % (p['WRITE_SIZE'] / 1024)) if 'WRITE_SIZE' in p else "",
                ("%.1f" % (p['SQ_INSTS_VALU'] / 1e6)) if 'SQ_INSTS_VALU' in p else "",
                ("%.2f" % issue) if issue is not None else ""))
        f.write("\nVALU floor frac = SQ_INSTS_VALU x 2.8 cycles / (1024 SIMDs x kernel time x 2.4 GHz): the kernel's vector instruction stream "
                "at the rate the chip retires independent f32 instructions with four waves per SIMD (scripts/ubench_valu.hip, "
                "profiles/r06_ubench_valu.txt: 5.8-6.3 cycles per wave-instruction with one wave per SIMD, 3.0-3.3 with two, 2.8 with four), "
                "over its measured time -- the roofline that applies to the render kernels (DESIGN.md section 4, round 6).  Earlier "
                "rounds' summaries priced an instruction at 2 cycles (the data-sheet rate): multiply their column by 1.4.\n")
        f.write("\nNotes: FETCH_SIZE on gfx950 under-reports wide (16 B/lane) streaming reads by 2x (MI355X_MICROARCH.md, HBM); the "
                "render kernels gather 16-byte records, so their true fetch lies between the reported value and twice it.  Kernels named "
                "bucket_* / minmax / scan_reduce / scan_apply belong to the un-hinted depth order (the first call for a given number of "
                "Gaussians, and every 64th call, which refreshes the depth-range hint); `__amd_rocclr_*` are the runtime's fill / copy "
                "kernels (torch tensor initialisation).  Counters of the single-view step with derived utilisations: "
                "r03a_single_view_summary.txt (round 3's kernels; the render / geometry-backward kernels are unchanged since).\n")
        v = os.path.join(P, "%s_voxel256_kernel_stats.csv" % tag)
        if os.path.exists(v):
            f.write("\n## Voxelizer alone: 256^3 query of the same cloud (`scripts/voxel_query_only.py 12`)\n\n| kernel | calls | avg us | % |\n|---|---|---|---|\n")
            for r in list(csv.DictReader(open(v)))[:16]:
                f.write("| `%s` | %s | %.1f | %s |\n" % (short(r['Name']), r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
    print(open(os.path.join(P, "%s_summary.md" % tag)).read()[:2500])


# ---- the 256^3 voxel query alone (counters of the final voxel kernels)
vs = os.path.join(P, "%s_voxel256_kernel_stats.csv" % tag)
vp = os.path.join(P, "%s_voxel256_pmc.json" % tag)
if os.path.exists(vs):
    vrows = list(csv.DictReader(open(vs)))
    vpmc = json.load(open(vp)) if os.path.exists(vp) else {}
    with open(os.path.join(P, "%s_voxel256_summary.md" % tag), "w") as f:
        f.write("# Voxelizer, 256^3 query of the 300k-Gaussian benchmark cloud alone (%s)\n\n" % tag)
        f.write("`rocprofv3 --kernel-trace --stats -- python scripts/voxel_query_only.py 12` (12 calls on the stick-first chain, "
                "csrc/voxel_sticks.hip) and separate `--pmc` passes over the same command (FETCH_SIZE, WRITE_SIZE, SQ_*), averaged per launch.\n\n")
        f.write("| kernel | calls | avg us | % | FETCH MB | WRITE MB | VALU inst (M) | waves |\n|---|---|---|---|---|---|---|---|\n")
        for r in vrows[:24]:
            k = short(r['Name']); p = vpmc.get(k, {})
            f.write("| `%s` | %s | %.1f | %s | %s | %s | %s | %s |\n" % (
                k, r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage'],
                ("%.1f" % (p['FETCH_SIZE'] / 1024)) if 'FETCH_SIZE' in p else "",
                ("%.1f" % (p['WRITE_SIZE'] / 1024)) if 'WRITE_SIZE' in p else "",
                ("%.1f" % (p['SQ_INSTS_VALU'] / 1e6)) if 'SQ_INSTS_VALU' in p else "",
                ("%d" % p['SQ_WAVES']) if 'SQ_WAVES' in p else ""))
