#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -k "encoder or tiled or affine or fuzz or mixture_tok" > gpurun_out/s3_tests.txt 2>&1
echo "tests exit $?" >> gpurun_out/s3_tests.txt
timeout 600 python tools/sweep_nll.py > gpurun_out/s3_sweep_nll.txt 2>&1
timeout 600 python tools/encoder_probe.py > gpurun_out/r02_encoder_probe.txt 2>&1
timeout 600 python bench.py > gpurun_out/s3_bench.json 2> gpurun_out/s3_bench.err
tail -4 gpurun_out/s3_tests.txt
cat gpurun_out/s3_sweep_nll.txt
tail -12 gpurun_out/r02_encoder_probe.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/s3_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], json.dumps(d["roofline"], indent=1))
PY
