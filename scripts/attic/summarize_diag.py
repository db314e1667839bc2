"""Joins the rocprofv3 passes of scripts/gpu_r3_diag.sh into one per-kernel table (means per launch):
duration (un-countered kernel trace), SQ / GRBM counters, derived: effective clock = GRBM_GUI_ACTIVE / duration of the
counter pass, VALU issue utilisation = SQ_ACTIVE_INST_VALU*4 / (SIMDs * busy cycles) (SQ_* cycle counters tick in
quad-cycles, MI355X_MICROARCH.md "rocprofv3 PMC slots"), HBM bytes (FETCH_SIZE / WRITE_SIZE in KB on gfx950).
    python scripts/summarize_diag.py gpurun_out/diag_TAG [--json out.json]
"""
import collections
import csv
import glob
import json
import re
import sys


def short(n):
    m = re.search(r'(r2::(?:\(anonymous namespace\)::)?[a-zA-Z_0-9]+(?:<[^>(]*>)?)', n)
    return m.group(1).replace('(anonymous namespace)::', '').replace('r2::', '') if m else n.split('(')[0][:50]


def main():
    root = sys.argv[1]
    res = collections.defaultdict(dict)
    # un-countered kernel trace: durations
    for f in glob.glob(root + "/kt/**/*kernel_trace.csv", recursive=True):
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in dur.items():
            v = v[len(v) // 5:]   # drop warm-up launches
            res[k]["us"] = sum(v) / len(v)
            res[k]["launches"] = len(v)
    for d in sorted(glob.glob(root + "/pmc*/")):
        fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if not fs:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(fs[0])):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for fld in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size"):
                if fld in r:
                    res[k][fld] = r[fld]
        # durations inside this counter pass (kernels serialised): for the clock estimate
        kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
        pdur = collections.defaultdict(list)
        if kt:
            for r in csv.DictReader(open(kt[0])):
                pdur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, cs in agg.items():
            for c, v in cs.items():
                v = v[len(v) // 5:]
                res[k][c] = sum(v) / len(v)
            if "GRBM_GUI_ACTIVE" in cs and pdur.get(k):
                v = pdur[k][len(pdur[k]) // 5:]
                res[k]["us_in_grbm_pass"] = sum(v) / len(v)
    SIMDS, XCDS = 1024, 8
    out = {}
    for k, d in res.items():
        if "us" not in d:
            continue
        if d.get("GRBM_GUI_ACTIVE") and d.get("us_in_grbm_pass"):
            d["clock_ghz"] = d["GRBM_GUI_ACTIVE"] / XCDS / d["us_in_grbm_pass"] / 1e3   # the counter is summed over the 8 XCDs
        clk = d.get("clock_ghz", 2.4)
        if d.get("SQ_ACTIVE_INST_VALU") is not None:
            # quad-cycles of VALU issue summed over SIMDs / (SIMDs x kernel cycles)
            d["valu_util"] = d["SQ_ACTIVE_INST_VALU"] * 4 / (SIMDS * d["us"] * clk * 1e3)
        if d.get("SQ_INSTS_VALU") is not None:
            d["valu_util_at_2cyc"] = d["SQ_INSTS_VALU"] * 2 / (SIMDS * d["us"] * clk * 1e3)
        if d.get("FETCH_SIZE") is not None and d.get("WRITE_SIZE") is not None:
            d["hbm_MB"] = (d["FETCH_SIZE"] + d["WRITE_SIZE"]) / 1e3   # KB -> MB
        out[k] = d
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1, sort_keys=True)
    # per-step time: kernels of the steady-state step are launched once per step; the un-hinted fallback kernels only now and then
    nstep = max(d["launches"] for d in out.values())
    for d in out.values():
        d["us_per_step"] = d["us"] * d["launches"] / nstep
    tot = sum(d["us_per_step"] for d in out.values())
    print("kernel (single-view step)                          us   share   VGPR  clock  VALUinsts  util(ACTIVE) util(2cyc)  LDSinsts  bankconf   WAIT_INST  WAIT_ANY  fetchMB writeMB")
    for k, d in sorted(out.items(), key=lambda kv: -kv[1]["us"]):
        g = lambda c, s=1.0: ("%9.3g" % (d[c] * s)) if d.get(c) is not None else "        -"
        print("%-46s %7.2f %6.1f%% %5s %6s %9s %9s %9s %9s %9s %9s %9s %8s %8s" % (
            k[:46], d["us"], 100 * d["us_per_step"] / tot, d.get("VGPR_Count", "-"), ("%.2f" % d["clock_ghz"]) if "clock_ghz" in d else "-",
            g("SQ_INSTS_VALU"), ("%.3f" % d["valu_util"]) if "valu_util" in d else "-",
            ("%.3f" % d["valu_util_at_2cyc"]) if "valu_util_at_2cyc" in d else "-", g("SQ_INSTS_LDS"), g("SQ_LDS_BANK_CONFLICT"),
            g("SQ_WAIT_INST_ANY"), g("SQ_WAIT_ANY"), g("FETCH_SIZE", 1e-3), g("WRITE_SIZE", 1e-3)))
    print("sum of kernel time per step: %.1f us  (quad-cycle SQ counters; util(ACTIVE) = SQ_ACTIVE_INST_VALU*4 / (1024 SIMDs x kernel cycles),\n"
          " util(2cyc) = SQ_INSTS_VALU*2 / the same: the share of the VALU issue slots used if an instruction took the guide's 2 cycles)" % tot)


if __name__ == "__main__":
    main()
