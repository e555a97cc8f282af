"""Basic blocks of one kernel in a gfx950 .s file with instruction-class counts (fp64 VALU / other VALU / SALU / LDS / memory) and
their branch targets — to find what a loop body or a tail really executes.  Usage: isa_blocks.py file.s <mangled-name> [min_instrs]"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 1
st = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
en = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
bb, blocks, order, notes = "entry", {"entry": Counter()}, ["entry"], {}
for l in lines[st + 1:en]:
    m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
    if m:
        bb = m.group(1); blocks[bb] = Counter(); order.append(bb); notes[bb] = m.group(2).strip(); continue
    m = re.match(r"^; %bb\.(\d+):(.*)", l)
    if m:
        bb = "bb." + m.group(1); blocks[bb] = Counter(); order.append(bb); notes[bb] = m.group(2).strip(); continue
    m = re.match(r"\s+([a-z_0-9]+)", l)
    if not m:
        continue
    op = m.group(1)
    if "_f64" in op and op.startswith("v_"): k = "f64"
    elif op.startswith("v_"): k = "valu"
    elif op.startswith("s_"): k = "salu"
    elif op.startswith("ds_"): k = "lds"
    else: k = "mem"
    blocks[bb][k] += 1
    if "branch" in op: blocks[bb]["->" + l.split()[-1]] += 1
for b in order:
    c = blocks[b]
    n = sum(v for k, v in c.items() if not k.startswith("->"))
    if n >= thr:
        print("%-12s f64 %4d valu %4d salu %4d lds %3d mem %3d  %s  %s" % (b, c["f64"], c["valu"], c["salu"], c["lds"], c["mem"],
              " ".join(k for k in c if k.startswith("->")), notes.get(b, "")[:70]))
