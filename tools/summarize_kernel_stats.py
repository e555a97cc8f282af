"""Top kernels of a rocprofv3 --kernel-trace --stats run and the share of this library's kernels (namespace cnf::).
Usage: summarize_kernel_stats.py <kernel_stats.csv> <out.csv> "<what was run>" [rows]"""
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
head, body = rows[0], rows[1:]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 30
tot = sum(float(r[2]) for r in body)
ours = sum(float(r[2]) for r in body if "cnf::" in r[0])
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["# %s: total kernel time %.1f ms, kernels of this library (cnf::) %.2f %% of it" % (sys.argv[3], tot / 1e6, 100.0 * ours / tot)])
    w.writerows([head] + [[r[0][:110]] + r[1:] for r in body[:n]])
print("total %.1f ms, cnf:: %.2f %%" % (tot / 1e6, 100.0 * ours / tot))
for r in body[:12]:
    print("%6.2f %%  %8.1f us avg  x%-6s %s" % (float(r[4]), float(r[3]) / 1e3, r[1], r[0][:100]))
