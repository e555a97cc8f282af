"""fp64 VALU ceiling of the reference-precision mixture kernels from the passes of tools/pmc_fp64.sh.  Issue costs per
wave-instruction and SIMD, measured (tools/microbench/op_rates.hip, profiles/r05_op_rates.txt): v_fma / v_mul / v_add_f64 4 cycles
(= the 78.6 TFLOP/s fp64 vector peak of MI355X_MICROARCH.md: 256 CUs x 4 SIMDs x 16 lanes x 2 flops x 2.4 GHz), v_rcp / v_rsq_f64
16, fp32 transcendentals 8, everything else (conversions, integer and fp32 plain instructions) between 2 and 4: two bounds.
Usage: fp64_ceilings.py <dir>"""
import csv, glob, json, os, sys
from collections import defaultdict
d = sys.argv[1]
man = json.load(open(os.path.join(d, "manifest.json")))
rows = defaultdict(lambda: defaultdict(list))          # (kernel, dispatch order) -> counter -> values
for path in glob.glob(os.path.join(d, "pass*", "**", "*counter_collection.csv"), recursive=True):
    per_kernel = defaultdict(list)
    for r in csv.DictReader(open(path)):
        per_kernel[(r["Kernel_Name"], r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["Grid_Size"])))
    for (k, cn), lst in per_kernel.items():
        for did, v, gs in sorted(lst):
            rows[(k, gs)][cn].append(v)
dur = defaultdict(list)
for path in glob.glob(os.path.join(d, "stats", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        dur[(r["Kernel_Name"], int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)))].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
print("%-64s %9s %9s %9s %9s %9s %9s %11s %11s %8s %8s" % ("kernel (grid)", "us", "fma64", "mul64", "add64", "trans64", "other", "fp64 2-cyc", "fp64 4-cyc", "wait %", "hbm frac"))
out = []
for (k, gs), c in sorted(rows.items()):
    if "mixture" not in k:
        continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    us = None
    for (kk, g2), lst in dur.items():
        if kk == k and (g2 == gs or g2 * 1 == gs):
            us = sum(lst) / len(lst) / 1e3
    fma, mul, add, t64, t32, valu = (m.get("SQ_INSTS_VALU_" + n, 0.0) for n in ("FMA_F64", "MUL_F64", "ADD_F64", "TRANS_F64", "TRANS_F32", ""))
    valu = m.get("SQ_INSTS_VALU", 0.0)
    other = valu - fma - mul - add - t64 - t32
    simd = 128.0 * m.get("GRBM_GUI_ACTIVE", 0.0)
    lo = ((fma + mul + add) * 4 + t64 * 16 + t32 * 8 + other * 2) / simd if simd else float("nan")
    hi = ((fma + mul + add) * 4 + t64 * 16 + t32 * 8 + other * 4) / simd if simd else float("nan")
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    wait = 100.0 * m.get("SQ_WAIT_ANY", 0.0) / wc if wc else float("nan")
    name = k.replace("void cnf::", "").split("(")[0]
    print("%-64s %9s %9.3g %9.3g %9.3g %9.3g %9.3g %11.2f %11.2f %8.0f" % (("%s (%d)" % (name, gs))[:64], "%.1f" % us if us else "-", fma, mul, add, t64, other, lo, hi, wait))
    extra = {n: m[n] for n in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_SCA",
                               "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_VALU_CVT",
                               "SQ_INSTS_VALU_INT32", "GRBM_GUI_ACTIVE", "SQ_WAVES") if n in m}
    if extra:
        print("      " + "  ".join("%s %.3g" % (n.replace("SQ_", ""), v) for n, v in extra.items()))
    out.append({"kernel": name, "grid": gs, "us": us, "fma_f64": fma, "mul_f64": mul, "add_f64": add, "trans_f64": t64, "trans_f32": t32, "other_valu": other,
                "fp64_issue_share_low": lo, "fp64_issue_share_high": hi, "wait_pct": wait})
json.dump({"manifest": man, "kernels": out}, open(os.path.join(d, "fp64_ceilings.json"), "w"), indent=1)
