"""The encoder backward's kernels, one row each, from the rocprofv3 passes of tools/pmc_passes.sh over
tools/pmc_encoder_bwd_workload.py (VERDICT r4 #1a: what binds them):
    bash tools/pmc_passes.sh pmc_encbwd_c16 python tools/pmc_encoder_bwd_workload.py 16 1,12
    bash tools/pmc_passes.sh pmc_encbwd_c51 python tools/pmc_encoder_bwd_workload.py 51 1,12
    python tools/encoder_bwd_breakdown.py gpurun_out/pmc_encbwd_c16:16 gpurun_out/pmc_encbwd_c51:51 > profiles/r05_encoder_bwd_breakdown.txt
Columns: us per launch (rocprofv3 --stats); VALU and transcendental instructions per (token, class) pair and wave ("slot":
64 pairs); the VALU issue share of the SIMD cycles of the launch under the two bounds the counters allow (every plain
instruction at the 2-cycle rate of v_fma / v_mul / v_add, or at the 4-cycle rate of v_bfi / v_cndmask / v_cmp / packed f32 —
profiles/r05_op_rates.txt; transcendentals 8 cycles): the true share lies between them; s_waitcnt stalls and issue stalls in
percent of the waves' resident cycles; LDS bank-conflict cycles per LDS-active cycle; mean resident waves per SIMD."""
import csv, glob, os, sys
from collections import defaultdict

T, D = 16384 * 64, 6
print(__doc__.split("Columns:")[1].strip().replace("\n", " "))
print()
print("%-4s %-44s %8s %9s %9s %11s %11s %8s %8s %9s %8s" % ("C", "kernel", "us", "VALU/slot", "trans/slot", "valu 2-cyc", "valu 4-cyc", "waitcnt%", "issue %", "LDS confl", "waves/SIMD"))
for arg in sys.argv[1:]:
    ds, C = arg.rsplit(":", 1)          # dir[+dir...]:C (counter passes of one workload may sit in several directories)
    C = int(C)
    vals = defaultdict(lambda: defaultdict(list))
    dur = {}
    for d in ds.split("+"):
        for path in glob.glob(os.path.join(d, "pass*", "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                vals[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for path in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                dur[row["Name"]] = float(row["AverageNs"])
    slots = T / 64.0 * C
    total = defaultdict(float)
    for k in sorted(vals):
        if "encoder_bwd" not in k:
            continue
        c = {n: sum(v) / len(v) for n, v in vals[k].items()}
        us = dur.get(k, float("nan")) / 1e3
        valu, trans, gui = c.get("SQ_INSTS_VALU", 0.0), c.get("SQ_INSTS_VALU_TRANS_F32", float("nan")), c.get("GRBM_GUI_ACTIVE", 0.0)
        simd_cycles = 128.0 * gui                      # GRBM_GUI_ACTIVE sums the 8 XCDs, 128 SIMDs each
        f2 = ((valu - trans) * 2 + trans * 8) / simd_cycles if simd_cycles else float("nan")
        f4 = ((valu - trans) * 4 + trans * 8) / simd_cycles if simd_cycles else float("nan")
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        name = k.replace("void cnf::", "").split("(")[0]
        per_slot = slots if ("token" in name or "class" in name or "pairs" in name) else float("nan")
        print("%-4d %-44s %8.1f %9.1f %9.1f %11.2f %11.2f %8.0f %8.0f %9.2f %8.1f" % (
            C, name[:44], us, valu / per_slot, trans / per_slot, f2, f4,
            100.0 * c.get("SQ_WAIT_ANY", 0.0) / wc if wc else float("nan"), 100.0 * c.get("SQ_WAIT_INST_ANY", 0.0) / wc if wc else float("nan"),
            c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0),
            wc * 4.0 / simd_cycles if simd_cycles else float("nan")))
    print()
