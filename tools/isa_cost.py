"""Issue-cost estimate of a range of gfx950 ISA lines from the opcode classes measured by tools/microbench/op_rates.hip
(profiles/r05_op_rates.txt): 2-cycle VALU (fma / mul / add / sub / and / or / xor / mov / add_u32) 1.05 ns, 4-cycle VALU
(bfi, min / max, cndmask, cmp, dpp, shifts, mad_u24, cvt, packed f32) 1.76 ns, transcendental 3.43 ns per wave-instruction
per SIMD.  Usage: isa_cost.py file.s <mangled-name-fragment> [first_line last_line]  (line numbers relative to the kernel label)"""
import re, sys
from collections import Counter
path, frag = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and frag in l and l.rstrip().endswith(":") is False and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, end - start)
FAST = ("v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fmac_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32",
        "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mul_legacy_f32", "v_accvgpr", "v_mov_b64", "v_add_co_u32", "v_addc_co_u32", "v_not_b32")
TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32")
cnt, ops = Counter(), Counter()
for l in lines[start + lo:start + hi]:
    m = re.match(r"\s+([a-z_0-9]+)", l)
    if not m:
        continue
    op = m.group(1)
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if op.startswith("v_"):
        if base in TRANS: k = "trans"
        elif op.endswith("_dpp"): k = "valu4"
        elif base in FAST: k = "valu2"
        else: k = "valu4"
    elif op.startswith("ds_"): k = "lds"
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): k = "vmem"
    elif op.startswith("s_"): k = "salu"
    else: k = "other"
    cnt[k] += 1
    ops[(k, base)] += 1
ns = cnt["valu2"] * 1.05 + cnt["valu4"] * 1.76 + cnt["trans"] * 3.43
print("lines %d..%d of %s: %s" % (lo, hi, frag, dict(cnt)))
print("VALU issue cost %.1f ns per wave (2-cycle %d, 4-cycle %d, transcendental %d)" % (ns, cnt["valu2"], cnt["valu4"], cnt["trans"]))
for (k, o), n in sorted(ops.items(), key=lambda kv: -kv[1])[:40]:
    print("   %-6s %-24s %d" % (k, o, n))
