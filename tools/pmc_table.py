"""Per-kernel means of every counter found under <dir>/pass*/ (rocprofv3 --pmc CSVs) plus the --stats durations.
Usage: pmc_table.py <dir>"""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]
vals = defaultdict(lambda: defaultdict(list))
meta = {}
for path in glob.glob(os.path.join(d, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if "cnf::" not in k and "copyBuffer" not in k and "reader<" not in k:
            continue
        vals[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        meta[k] = (row.get("VGPR_Count") or row.get("Arch_VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"),
                   row.get("Workgroup_Size"), row.get("Grid_Size"))
dur = {}
for path in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        dur[row["Name"]] = (float(row["AverageNs"]), int(row["Calls"]))
for k in sorted(vals):
    print("==", k[:150])
    print("   vgpr/sgpr/lds/wg/grid:", meta[k], " avg_ns,calls:", dur.get(k))
    c = {n: sum(v) / len(v) for n, v in vals[k].items()}
    for n in sorted(c):
        print("   %-24s %16.1f  (n=%d)" % (n, c[n], len(vals[k][n])))
    if c.get("SQ_WAVE_CYCLES"):
        wc = c["SQ_WAVE_CYCLES"]
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if n in c:
                print("   %-24s %6.1f %% of wave cycles" % (n, 100.0 * c[n] / wc))
    if c.get("SQ_BUSY_CYCLES") and c.get("SQ_WAVE_CYCLES"):
        print("   mean resident waves per SQ-busy cycle: %.2f" % (c["SQ_WAVE_CYCLES"] / c["SQ_BUSY_CYCLES"]))
    if "FETCH_SIZE" in c:
        print("   read  %.2f MB (FETCH_SIZE x2, gfx950)   write %s MB" % (c["FETCH_SIZE"] * 2 * 1024 / 1e6,
              ("%.2f" % (c["WRITE_SIZE"] * 1024 / 1e6)) if "WRITE_SIZE" in c else "?"))
