"""Whole-flow HBM traffic from the passes of tools/pmc_passes.sh over tools/flow_traffic_workload.py {fused,unfused}:
bytes moved by this library's kernels per evaluation pass and per flow step, next to the algorithmic figures of
SURVEY.md 8d.  Usage: flow_traffic.py <fused dir> <unfused dir> [out.json]"""
import csv, glob, json, os, sys
from collections import defaultdict

B, N, D, K, STEPS, REP = 16384, 16, 4, 8, 8, 6
elems = B * N * D


def load(d):
    vals = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(d, "pass*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if "cnf::" in row["Kernel_Name"] and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                vals[row["Kernel_Name"].split("(")[0][:90]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    dur = {}
    for path in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if "cnf::" in row["Name"]:
                dur[row["Name"].split("(")[0][:90]] = (float(row["TotalDurationNs"]), int(row["Calls"]))
    out, total_b, total_ns = [], 0.0, 0.0
    for k, c in vals.items():
        rd = sum(c.get("FETCH_SIZE", [])) * 2 * 1024          # gfx950: FETCH_SIZE counts half of a wide coalesced read
        wr = sum(c.get("WRITE_SIZE", [])) * 1024
        n = max(len(c.get("FETCH_SIZE", [])), len(c.get("WRITE_SIZE", [])))
        t = dur.get(k, (0.0, 0))
        out.append({"kernel": k, "launches_per_pass": n / REP, "read_MB_per_pass": rd / REP / 1e6, "write_MB_per_pass": wr / REP / 1e6,
                    "us_per_pass": t[0] / REP / 1e3})
        total_b += (rd + wr) / REP
        total_ns += t[0] / REP
    return out, total_b, total_ns


res = {}
for tag, d in (("fused", sys.argv[1]), ("unfused", sys.argv[2])):
    rows, tb, tns = load(d)
    res[tag] = {"kernels": rows, "MB_per_pass": tb / 1e6, "bytes_per_elem_per_flow_step": tb / STEPS / elems, "kernel_us_per_pass": tns / 1e3}
    print("== %s: %.1f MB per pass = %.1f B/elem per flow step, %.1f us of kernels per pass" % (tag, tb / 1e6, tb / STEPS / elems, tns / 1e3))
    for r in sorted(rows, key=lambda r: -r["read_MB_per_pass"] - r["write_MB_per_pass"]):
        print("   %-72s x%4.1f  read %8.1f MB  write %7.1f MB  %7.1f us" % (r["kernel"][-72:], r["launches_per_pass"], r["read_MB_per_pass"],
                                                                      r["write_MB_per_pass"], r["us_per_pass"]))
alg_step = 16 + 12 * K
res["algorithmic"] = {"coupling_B_per_elem": alg_step, "unfused_step": alg_step + 16, "actnorm_conv_fused_step": alg_step + 8,
                      "three_way_fused_step": alg_step, "note": "SURVEY.md 8d: mixture coupling 16 + 12K (all parameter blocks counted), ActNorm and 1x1 conv 8 each, "
                      "ActNorm + conv in one pass 8, coupling + next step's ActNorm + conv in one pass 16 + 12K; prior / NLL 4 more on the last step unless fused"}
print(json.dumps(res["algorithmic"]))
if len(sys.argv) > 3:
    json.dump(res, open(sys.argv[3], "w"), indent=1)
