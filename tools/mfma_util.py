"""MFMA utilisation per GEMM kernel from the passes of tools/mfma_util.sh.
  flops      = SQ_INSTS_VALU_MFMA_MOPS_<dtype> * 512            (one MOPS unit = 512 FLOPs; rocprof's MfmaFlops* definition)
  util_flops = flops / duration / dense peak of the dtype        (MI355X_MICROARCH.md: f32-input MFMA 157.3 TFLOP/s, bf16/f16 2500)
  util_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * SQ_BUSY_CU_CYCLES)     (share of busy-CU SIMD cycles the matrix pipe is occupied)
Usage: mfma_util.py <dir> [out.json]"""
import csv, glob, json, os, sys
from collections import defaultdict

d = sys.argv[1]
PEAK = {"F32": 157.3e12, "BF16": 2500e12, "F16": 2500e12}
vals = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
for path in glob.glob(os.path.join(d, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        vals[row["Kernel_Name"]][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
            calls[row["Kernel_Name"]] += 1
dur = {}
total_ns = 0.0
for path in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        dur[row["Name"]] = (float(row["TotalDurationNs"]), int(row["Calls"]))
        total_ns += float(row["TotalDurationNs"])
rows = []
tot_flops = 0.0
for k, c in vals.items():
    mops = {t: c.get("SQ_INSTS_VALU_MFMA_MOPS_" + t, 0.0) for t in PEAK}
    if sum(mops.values()) <= 0 or k not in dur:
        continue
    t_ns, n = dur[k]
    scale = n / max(calls[k], 1)                      # the stats run and the counter run launch the same kernels
    dtype = max(mops, key=mops.get)
    flops = mops[dtype] * 512.0 * scale
    tot_flops += flops
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * c["SQ_BUSY_CU_CYCLES"]) if c.get("SQ_BUSY_CU_CYCLES") else None
    rows.append({"kernel": k[:100], "calls": n, "total_ms": t_ns / 1e6, "share_of_gpu_time": t_ns / total_ns, "dtype": dtype,
                 "TFLOPs": flops / t_ns / 1e3, "mfma_util_vs_peak": flops / (t_ns * 1e-9) / PEAK[dtype], "mfma_pipe_busy": busy})
rows.sort(key=lambda r: -r["total_ms"])
print("%-70s %6s %9s %7s %5s %8s %9s %9s" % ("GEMM kernel (MFMA)", "calls", "total ms", "share", "dtype", "TFLOP/s", "of peak", "pipe busy"))
for r in rows[:25]:
    print("%-70s %6d %9.2f %6.1f%% %5s %8.1f %8.1f%% %9s" % (r["kernel"][:70], r["calls"], r["total_ms"], 100 * r["share_of_gpu_time"], r["dtype"],
          r["TFLOPs"], 100 * r["mfma_util_vs_peak"], "%.1f%%" % (100 * r["mfma_pipe_busy"]) if r["mfma_pipe_busy"] is not None else "-"))
mfma_ms = sum(r["total_ms"] for r in rows)
summary = {"gpu_time_ms": total_ns / 1e6, "mfma_kernels_ms": mfma_ms, "mfma_kernels_share": mfma_ms / (total_ns / 1e6) if total_ns else None,
           "mfma_TFLOPs_over_mfma_kernel_time": tot_flops / (mfma_ms * 1e-3) / 1e12 if mfma_ms else None,
           "mfma_util_vs_f32_peak_over_mfma_kernel_time": tot_flops / (mfma_ms * 1e-3) / PEAK["F32"] if mfma_ms else None,
           "mfma_util_vs_f32_peak_over_whole_step": tot_flops / (total_ns * 1e-9) / PEAK["F32"] if total_ns else None}
print(json.dumps(summary))
if len(sys.argv) > 2:
    json.dump({"summary": summary, "kernels": rows[:40]}, open(sys.argv[2], "w"), indent=1)
