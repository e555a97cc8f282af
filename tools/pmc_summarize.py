"""Turn the rocprofv3 --pmc counter_collection CSVs into per-launch HBM traffic, calibrated on the
known-size copy in the same pass.  Usage: pmc_summarize.py <fetch_dir> <write_dir> <out_json> <out_txt>"""
import csv, glob, json, os, sys
from collections import defaultdict


def collect(d, counter):
    vals = defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == counter:
                vals[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return vals


def pick(vals, key):
    out = []
    for k, v in vals.items():
        if key in k:
            out += v
    return out


fetch = collect(sys.argv[1], "FETCH_SIZE")
write = collect(sys.argv[2], "WRITE_SIZE")
COPY_BYTES = 16384 * 64 * 6 * 4 * 4.0         # bytes read (= bytes written) per calibration copy launch
# template arguments: <VEC, U, HAS_SF, REVERSE, FAST, NLL>; "affine_coupling_fwd" is the kernel bench.py's roofline names
names = {"affine_coupling_fwd": "affine_coupling_kernel<4, 2, true, false, true, 1>",
         "affine_coupling_fwd_plain": "affine_coupling_kernel<4, 2, true, false, true, 0>",
         "affine_coupling_inv": "affine_coupling_kernel<4, 2, true, true, true, 0>",
         "mixture_fwd": "mixture_tok_kernel<8, false, 1, false, 0>", "copy": "__amd_rocclr_copyBuffer"}
med = lambda x: sorted(x)[len(x) // 2] if x else None
big = lambda x: [v for v in x if v > 0.5 * max(x)] if x else x      # the calibration copies, not the tiny H2D/D2H ones
raw = {}
for tag, key in names.items():
    f, w = pick(fetch, key), pick(write, key)
    if tag == "copy":
        f, w = big(f), big(w)
    raw[tag] = {"FETCH_SIZE_KB": med(f), "WRITE_SIZE_KB": med(w), "launches": len(f)}
cf = COPY_BYTES / (raw["copy"]["FETCH_SIZE_KB"] * 1024.0) if raw["copy"]["FETCH_SIZE_KB"] else None
cw = COPY_BYTES / (raw["copy"]["WRITE_SIZE_KB"] * 1024.0) if raw["copy"]["WRITE_SIZE_KB"] else None
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_sources_sha
out = {"kernel_sources_sha": kernel_sources_sha(),
       "calibration": {"copy_bytes_each_way": COPY_BYTES, "fetch_factor": cf, "write_factor": cw,
                       "note": "factor = known bytes / (counter KB * 1024) on the 100.66 MB d2d copy (__amd_rocclr_copyBuffer) of the "
                               "same pass; MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of a wide coalesced read on gfx950"},
       "raw_median_per_launch": raw}
for tag in ("affine_coupling_fwd", "affine_coupling_fwd_plain", "affine_coupling_inv", "mixture_fwd"):
    f, w = raw[tag]["FETCH_SIZE_KB"], raw[tag]["WRITE_SIZE_KB"]
    if f is not None and w is not None and cf and cw:
        out[tag + "_read_bytes_per_launch"] = f * 1024.0 * cf
        out[tag + "_write_bytes_per_launch"] = w * 1024.0 * cw
        out[tag + "_bytes_per_launch"] = f * 1024.0 * cf + w * 1024.0 * cw
json.dump(out, open(sys.argv[3], "w"), indent=1)
with open(sys.argv[4], "w") as fh:
    fh.write(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
