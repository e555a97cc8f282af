"""Which ceiling binds each hot kernel: fraction of the 8 TB/s HBM peak (algorithmic bytes / duration) next to the
fraction of the fp32 VALU issue ceiling.  Inputs: the rocprofv3 passes of tools/pmc_ceilings.sh over
tools/pmc_ceilings_workload.py.  VALU ceiling: a wave64 VALU instruction occupies its SIMD-32 for 2 cycles, a
transcendental (v_exp / v_log / v_rcp) for 8 (quarter rate); MI355X has 1024 SIMDs, so
    valu_frac = ((INSTS_VALU - TRANS) * 2 + TRANS * 8) / (128 * GRBM_GUI_ACTIVE)
(MI355X_MICROARCH.md: v_fma_f32 wave64 = 2 cycles on a SIMD-32; GRBM_GUI_ACTIVE is summed over the 8 XCDs, each with
128 SIMDs: for a 136 us launch it reads 2.73e6 = 8 x 2.5 GHz x 136 us).
Round 3 calibrated both columns on the device (tools/microbench, profiles/r03_valu_calibration.txt): a SIMD saturated with
independent v_fma_f32 retires one per 2.0 cycles and v_exp / v_log / v_rcp one per 8 — the model above is what the
hardware does — while SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU is EXACTLY 1 quad-cycle for every plain and 2 for every
transcendental instruction at every occupancy: it is an instruction count in quantised units, not busy time.  Round 2's
`valu_busy` = SQ_ACTIVE_INST_VALU * 4 / SIMD cycles therefore over-states plain instructions 2x (1.73 for a saturated
SIMD).  It is still printed, as `quantised` (for the record), but `binds` is decided by `valu_frac` alone.
Usage: ceilings.py <pmc dir> [out.json]"""
import csv, glob, json, os, sys
from collections import defaultdict

d = sys.argv[1]
manifest = json.load(open(os.path.join(d, "manifest.json")))
vals = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(d, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        vals[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = {}
for path in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        dur[row["Name"]] = float(row["AverageNs"])
out = []
print("%-58s %9s %9s %9s %9s %9s %9s %9s %9s  %s" % ("kernel", "us", "alg GB/s", "hbm_frac", "valu_frac", "quantised", "cyc/inst", "wait %", "HBM MB", "binds"))
for frag, info in manifest.items():
    names = [k for k in vals if frag in k]
    if not names:
        continue
    k = names[0]
    c = {n: sum(v) / len(v) for n, v in vals[k].items()}
    t_ns = dur.get(k)
    if not t_ns:
        continue
    hbm = info["alg_bytes"] / t_ns / 8000.0
    trans = c.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
    valu = c.get("SQ_INSTS_VALU", 0.0)
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    valu_frac = ((valu - trans) * 2 + trans * 8) / (128.0 * gui) if gui else None
    act = c.get("SQ_ACTIVE_INST_VALU")
    valu_busy = act * 4.0 / (128.0 * gui) if (gui and act) else None
    cyc_inst = act * 4.0 / valu if (act and valu) else None
    wait = 100.0 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None
    mb = (c.get("FETCH_SIZE", 0) * 2 + c.get("WRITE_SIZE", 0)) * 1024 / 1e6 if "FETCH_SIZE" in c else None
    binds = "hbm" if hbm >= (valu_frac or 0.0) else "valu"
    row = {"kernel": frag, "what": info["what"], "avg_us": t_ns / 1e3, "alg_GBps": info["alg_bytes"] / t_ns, "hbm_frac": hbm,
           "valu_frac": valu_frac, "sq_active_inst_valu_x4_over_simd_cycles": valu_busy, "cycles_per_valu_inst": cyc_inst, "wave_wait_pct": wait, "hbm_MB_per_launch": mb, "valu_insts": valu, "trans_insts": trans,
           "clock_GHz_if_unprofiled_duration": gui / 8.0 / t_ns if gui else None, "binds": binds}
    out.append(row)
    print("%-58s %9.1f %9.0f %9.3f %9s %9s %9s %9s %9s  %s" % (info["what"][:58], t_ns / 1e3, info["alg_bytes"] / t_ns, hbm,
          "%.3f" % valu_frac if valu_frac is not None else "-", "%.3f" % valu_busy if valu_busy is not None else "-",
          "%.1f" % cyc_inst if cyc_inst is not None else "-", "%.0f" % wait if wait is not None else "-",
          "%.1f" % mb if mb is not None else "-", binds))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
