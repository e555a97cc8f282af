"""SQ counters of the calibration kernels of enc_micro (known instruction streams) -> what the derived figures read at a known
issue rate.  valu_busy = SQ_ACTIVE_INST_VALU x 4 / (128 SIMDs per XCD x GRBM_GUI_ACTIVE summed over 8 XCDs)."""
import csv, glob, os, sys
from collections import defaultdict, OrderedDict

rows = OrderedDict()
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["Grid_Size"]))
        rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
agg = defaultdict(lambda: defaultdict(list))
for (did, name, grid), c in rows.items():
    for k, v in c.items():
        agg[(name, grid)][k].append(v)
print("%-36s %6s %12s %12s %10s %10s %10s %10s" % ("kernel", "w/SIMD", "INSTS_VALU", "ACTIVE_VALU", "cyc/inst", "valu_busy", "wave_cyc/inst", "trans"))
for (name, grid), c in agg.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    wps = grid // 256 // 256
    iv, av, gui, wc = m.get("SQ_INSTS_VALU", 0), m.get("SQ_ACTIVE_INST_VALU", 0), m.get("GRBM_GUI_ACTIVE", 0), m.get("SQ_WAVE_CYCLES", 0)
    print("%-36s %6d %12.0f %12.0f %10.2f %10.3f %10.2f %10.0f" % (name[:36], wps, iv, av, av * 4 / iv if iv else 0,
          av * 4 / (128 * gui) if gui else 0, wc * 4 / iv * 1.0 if iv else 0, m.get("SQ_INSTS_VALU_TRANS_F32", -1)))
