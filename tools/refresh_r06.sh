#!/bin/bash
# Round 6's measurements for profiles/: the bench line and its rocprofv3 kernel summary, the PMC traffic of the dominant kernel and of
# the mixture forward (both layouts, both shapes).   gpurun --timeout 1500 -- 'bash tools/refresh_r06.sh'   then   python tools/collect_profiles.py r06
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; cd "$ROOT"
rm -rf "$OUT/prof_bench" "$OUT/pmc_fetch" "$OUT/pmc_write"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python tools/pmc_workload.py > "$OUT/pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- python tools/pmc_workload.py > "$OUT/pmc_write.log" 2>&1
python tools/pmc_summarize.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/traffic.json" "$OUT/traffic.txt" | tail -30
rm -f "$OUT"/pmc_fetch/*kernel_trace.csv "$OUT"/pmc_write/*kernel_trace.csv
cp "$OUT/traffic.json" "$ROOT/profiles/traffic.json"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bench" -o bench -- python bench.py --no-cpu-baseline > "$OUT/bench_prof.log" 2>&1
tail -1 "$OUT/bench_prof.log" | cut -c1-300
rm -f "$OUT"/prof_bench/*kernel_trace.csv
timeout 500 python bench.py > "$OUT/bench.log" 2>&1; tail -1 "$OUT/bench.log" | cut -c1-400
find "$OUT" -name "*counter_collection.csv" -size +4M -delete
