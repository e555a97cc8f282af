"""FETCH_SIZE / WRITE_SIZE of the streaming mixture backward at S*, calibrated on the known-size copy of the same pass.
Usage: pmc_mixture_bwd_summarize.py <fetch_dir> <write_dir> <out_txt>"""
import csv, glob, os, sys


def collect(d, counter):
    rows = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == counter:
                rows.setdefault(row["Kernel_Name"], []).append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    return rows


med = lambda x: sorted(x)[len(x) // 2]
B, N, D, K = 16384, 64, 6, 8
P, DA = 2 + 3 * K, 3
COPY = B * N * D * 4 * 4.0
out = []
res = {}
for tag, d, counter in (("read", sys.argv[1], "FETCH_SIZE"), ("write", sys.argv[2], "WRITE_SIZE")):
    rows = collect(d, counter)
    copies = [v for k, vs in rows.items() if "copyBuffer" in k for _, v in vs]
    copies = [v for v in copies if v > 0.5 * max(copies)]
    factor = COPY / (med(copies) * 1024.0)
    ks = sorted(v for k, vs in rows.items() if "mixture_tok_bwd_kernel" in k for v in vs)
    half = len(ks) // 2
    res[tag] = (factor, med([v for _, v in ks[:half]]) * 1024.0 * factor, med([v for _, v in ks[half:]]) * 1024.0 * factor)
    out.append("%s: calibration factor %.3f (known %.2f MB copy / counter), %d + %d launches" % (counter, factor, COPY / 1e6, half, len(ks) - half))
tok = B * N
need_c = {"read": tok * (DA * P * 4 + 8 * D) + 4 * B, "write": tok * (DA * P * 4 + 4 * D)}
need_r = {"read": tok * (DA * P * 4 + 8 * D) + 4 * B, "write": tok * (D * P * 4 + 4 * D)}
for i, (name, need) in enumerate((("compact layout", need_c), ("reference layout", need_r))):
    rd, wr = res["read"][1 + i], res["write"][1 + i]
    out.append("S* mixture backward, %s (streaming kernel): read %.1f MB (needed %.1f, x%.2f), written %.1f MB (needed %.1f, x%.2f), total %.1f MB = x%.2f"
               % (name, rd / 1e6, need["read"] / 1e6, rd / need["read"], wr / 1e6, need["write"] / 1e6, wr / need["write"], (rd + wr) / 1e6,
                  (rd + wr) / (need["read"] + need["write"])))
open(sys.argv[3], "w").write("\n".join(out) + "\n")
print("\n".join(out))
