#!/bin/bash
# One GPU session: new tests first, then the probes and a bench run.  Outputs under gpurun_out/.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/s2_tests.txt 2>&1
echo "tests exit $?" >> gpurun_out/s2_tests.txt
timeout 600 python tools/encoder_probe.py > gpurun_out/r02_encoder_probe.txt 2>&1
timeout 300 python tools/sustained_probe.py > gpurun_out/r02_sustained_probe.txt 2>&1
timeout 600 python bench.py > gpurun_out/s2_bench.json 2> gpurun_out/s2_bench.err
tail -30 gpurun_out/s2_tests.txt
cat gpurun_out/r02_sustained_probe.txt
tail -8 gpurun_out/r02_encoder_probe.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/s2_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], json.dumps(d["roofline"], indent=1))
PY
