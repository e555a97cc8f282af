#!/bin/bash
# Kernel-only A/B of library variants / knob settings under rocprofv3 (two interleaved rounds):
#   tools/ab_kernel_stats.sh <out.txt> "<bwd_probe --only filter>" "main" "main|--static 3 --U 2" "abl1|--U 1" ...
# (main = categoricalnf_amd/lib/libcnf_hip.so, other names = categoricalnf_amd/lib/var_<name>.so; after | extra bwd_probe flags)
out=$1; filt=$2; shift 2
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
rm -f "$out"
for r in 1 2; do for spec in "$@"; do
    v=${spec%%|*}; extra=""; [ "$spec" != "$v" ] && extra=${spec#*|}
    d=gpurun_out/ab_ks_tmp; rm -rf $d
    if [ $v = main ]; then unset CNF_LIB_OVERRIDE; else export CNF_LIB_OVERRIDE=categoricalnf_amd/lib/var_$v.so; fi
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ks -- python tools/bwd_probe.py --reps 30 --only "$filt" $extra > /dev/null 2>&1
    echo "== round $r $spec" >> "$out"
    python - $d/ks_kernel_stats.csv >> "$out" <<'PY'
import csv, sys
for r in sorted(list(csv.reader(open(sys.argv[1])))[1:], key=lambda r: r[0]):
    if "cnf::" in r[0] and "stream_mix" not in r[0]:
        print("%9.2f us  x%-5s %s" % (float(r[3]) / 1e3, r[1], r[0][:110]))
PY
    rm -rf $d
done; done
unset CNF_LIB_OVERRIDE
cat "$out"
