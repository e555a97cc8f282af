#!/bin/bash
# The driver's scaling run, rehearsable: bench.py at N = 1, 2, 4, 8 ranks of ONE node back to back (one process per GPU over
# RCCL; bench.py starts its ranks itself), one JSON line each, then the weak-scaling table (value(N) / (N * value(1))).
#   tools/scale_sweep.sh [outdir] [N ...]                     real devices
#   SHARE_DEVICE=1 tools/scale_sweep.sh out 1 2 8             all ranks on cuda:0 over gloo (a 1-GPU box: exercises the N-rank
#                                                             code path, not the interconnect — efficiencies mean nothing there)
# Extra bench.py flags go in BENCH_FLAGS (default: the driver's short run).
set -u
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/scale}; shift || true
ns=${*:-1 2 4 8}
mkdir -p "$out"
flags=${BENCH_FLAGS:---steps 200 --warmup 20 --no-cpu-baseline --no-mixture --no-kernel-table}
share=""
[ -n "${SHARE_DEVICE:-}" ] && share="--share-device --backend gloo"
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > "$out/scale.jsonl"
for n in $ns; do
    timeout 900 python bench.py --gpus "$n" $flags $share > "$out/bench_n$n.log" 2> "$out/bench_n$n.err"
    rc=$?
    line=$(grep '^{' "$out/bench_n$n.log" | tail -1)
    if [ $rc -ne 0 ] || [ -z "$line" ]; then echo "N=$n: bench.py failed (rc $rc): $(tail -2 "$out/bench_n$n.err" | tr '\n' ' ')"; continue; fi
    echo "$line" >> "$out/scale.jsonl"
done
python - "$out/scale.jsonl" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
if not rows:
    raise SystemExit("no bench line was produced")
base = next((r for r in rows if r["n_gpus"] == 1), rows[0])
print("%4s %14s %10s %10s %12s %14s %s" % ("N", "elems/s", "ms/step", "efficiency", "allreduce us", "kernel us min", "kernel us max"))
table = []
for r in rows:
    k = r["roofline"].get("per_rank_kernel_ms") or [r["roofline"]["kernel_ms"]]
    eff = r["value"] / (r["n_gpus"] * base["value"] / base["n_gpus"])
    print("%4d %14.4g %10.4f %10.3f %12s %14.2f %.2f" % (r["n_gpus"], r["value"], r["ms_per_step"], eff,
          "-" if r.get("allreduce_latency_us") is None else "%.1f" % r["allreduce_latency_us"], min(k) * 1e3, max(k) * 1e3))
    table.append({"n_gpus": r["n_gpus"], "value": r["value"], "unit": r.get("unit"), "ms_per_step": r["ms_per_step"], "efficiency": eff,
                  "allreduce_latency_us": r.get("allreduce_latency_us"), "kernel_us_min": min(k) * 1e3, "kernel_us_max": max(k) * 1e3,
                  "backend": r.get("backend"), "share_device": bool(r.get("share_device"))})
# the same table for programs (weak scaling: per-GPU work fixed; efficiency = value(N) / (N x value(1)))
import os
json.dump({"metric": base.get("metric"), "scaling": "weak", "rows": table}, open(os.path.join(os.path.dirname(sys.argv[1]), "scale.json"), "w"), indent=1)
PY
