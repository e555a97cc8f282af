"""Analyse a rocprofv3 kernel trace of bench.py: durations and start-to-start intervals of the affine forward
kernel inside the timed steps and inside the trailing back-to-back block.  Usage: trace_tail.py <kernel_trace.csv>"""
import csv, sys
import numpy as np
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
st = np.array([int(r["Start_Timestamp"]) for r in rows], dtype=np.int64)
en = np.array([int(r["End_Timestamp"]) for r in rows], dtype=np.int64)
isf = np.array([("affine_coupling_kernel<4, 2, true, false, true, 1>" in n) for n in names])
idx = np.nonzero(isf)[0]
# back-to-back block: forward launches whose predecessor is also a forward launch
b2b = [i for i in idx if i > 0 and isf[i - 1]]
stp = [i for i in idx if i > 0 and not isf[i - 1]]
for tag, sel in (("in-step", stp[-2000:]), ("back-to-back", b2b[-480:])):
    sel = np.array(sel)
    d = (en[sel] - st[sel]) / 1e3
    gap = (st[sel] - en[sel - 1]) / 1e3
    print("%-13s n=%5d  duration us mean %.2f med %.2f p10 %.2f p90 %.2f | gap-before us mean %.2f med %.2f"
          % (tag, len(sel), d.mean(), np.median(d), np.percentile(d, 10), np.percentile(d, 90), gap.mean(), np.median(gap)))
