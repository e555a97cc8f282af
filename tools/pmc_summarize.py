"""Turn the rocprofv3 --pmc counter_collection CSVs into per-launch HBM traffic, calibrated on the
known-size copy in the same pass.  Usage: pmc_summarize.py <fetch_dir> <write_dir> <out_json> <out_txt>"""
import csv, glob, json, os, sys
from collections import defaultdict


def collect(d, counter, ordered=None):
    """kernel name -> counter values; `ordered` (dict) additionally receives (dispatch id, value) pairs per kernel name"""
    vals = defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == counter:
                vals[row["Kernel_Name"]].append(float(row["Counter_Value"]))
                if ordered is not None:
                    ordered.setdefault(row["Kernel_Name"], []).append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    return vals


def pick(vals, key):
    out = []
    for k, v in vals.items():
        if key in k:
            out += v
    return out


fetch_ord, write_ord = {}, {}
fetch = collect(sys.argv[1], "FETCH_SIZE", fetch_ord)
write = collect(sys.argv[2], "WRITE_SIZE", write_ord)
COPY_BYTES = 16384 * 64 * 6 * 4 * 4.0         # bytes read (= bytes written) per calibration copy launch
# template arguments: <VEC, U, HAS_SF, REVERSE, FAST, NLL, ED>; "affine_coupling_fwd" is the kernel bench.py's roofline names
MIX = "mixture_tok_kernel<8, false, 1, false, 0, false, false>"
names = {"affine_coupling_fwd": "affine_coupling_kernel<4, 2, true, false, true, 1, 0>",
         "affine_coupling_fwd_plain": "affine_coupling_kernel<4, 2, true, false, true, 0, 0>",
         "affine_coupling_inv": "affine_coupling_kernel<4, 2, true, true, true, 0, 0>",
         "copy": "__amd_rocclr_copyBuffer"}
# tools/pmc_workload.py launches the mixture forward in four groups of equal size, in this order
MIX_CASES = [("mixture_fwd", 16384, 16, 4, False), ("mixture_fwd_compact", 16384, 16, 4, True),
             ("mixture_fwd_Sstar", 16384, 64, 6, False), ("mixture_fwd_Sstar_compact", 16384, 64, 6, True)]
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
def groups(ordered):
    rows = sorted(v for k, vs in ordered.items() if MIX in k for v in vs)
    n = len(rows) // len(MIX_CASES)
    return [[v for _, v in rows[i * n:(i + 1) * n]] for i in range(len(MIX_CASES))] if n else [[] for _ in MIX_CASES]
for (tag, Bm, Nm, Dm, compact), f, w in zip(MIX_CASES, groups(fetch_ord), groups(write_ord)):
    Km, DAm = 8, Dm - Dm // 2
    Pm = (2 + 3 * Km) * 4
    raw[tag] = {"FETCH_SIZE_KB": med(f), "WRITE_SIZE_KB": med(w), "launches": len(f),
                "needed_bytes": Bm * Nm * (DAm * Pm + 8 * Dm) + 4 * Bm,
                "contract_bytes": Bm * Nm * Dm * (16 + 12 * Km), "layout": "compact" if compact else "reference"}
for tag in ["affine_coupling_fwd", "affine_coupling_fwd_plain", "affine_coupling_inv"] + [c[0] for c in MIX_CASES]:
    f, w = raw[tag]["FETCH_SIZE_KB"], raw[tag]["WRITE_SIZE_KB"]
    if f is not None and w is not None and cf and cw:
        out[tag + "_read_bytes_per_launch"] = f * 1024.0 * cf
        out[tag + "_write_bytes_per_launch"] = w * 1024.0 * cw
        out[tag + "_bytes_per_launch"] = f * 1024.0 * cf + w * 1024.0 * cw
        if "needed_bytes" in raw[tag]:
            out[tag + "_over_needed"] = out[tag + "_bytes_per_launch"] / raw[tag]["needed_bytes"]
json.dump(out, open(sys.argv[3], "w"), indent=1)
with open(sys.argv[4], "w") as fh:
    fh.write(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
