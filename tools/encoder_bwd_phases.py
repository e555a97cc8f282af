"""Phase timers of the encoder backward's pair kernel (a -DCNF_ENC_BWD_PROBE build: tools/build_variant.sh encprobe
-DCNF_ENC_BWD_PROBE, run with CNF_LIB_OVERRIDE=categoricalnf_amd/lib/var_encprobe.so): mean / max over the workgroups of the
microseconds each wave spends per phase of the stage loop, summed over the workgroup's stages.
python tools/encoder_bwd_phases.py [B,N,D,C ...] [--variant V]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import ops, _lib
from categoricalnf_amd.ops import _ptr, _stream, _launch
dev = torch.device("cuda:0")
lib = _lib.load()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
variant = int(sys.argv[sys.argv.index("--variant") + 1]) if "--variant" in sys.argv else 2
args = [a for a in args if a != str(variant) or "--variant" not in sys.argv]
SHAPES = tuple(tuple(int(v) for v in a.split(",")) for a in args if "," in a) or ((16384, 64, 6, 16), (16384, 64, 6, 51))
NAMES = ("P0 + prefetch issue", "barrier 1", "A (pairs)", "barrier 2", "cells", "barrier 3", "finish", "barrier 4", "B", "cold / loop")
NT = 512 if variant in (4, 5) else (320 if variant in (6, 7) else 256)
if variant >= 6:
    NAMES = ("X (pairs | token wave)", "barrier a", "Y (class sums, own adds)", "barrier b") + ("-",) * 6
for B, N, D, C in SHAPES:
    g = torch.Generator(device=dev).manual_seed(1)
    categ = torch.randint(0, C, (B, N), generator=g, device=dev)
    table = 0.5 * torch.randn(C, 2 * D, generator=g, device=dev)
    prior = torch.log_softmax(torch.randn(C, generator=g, device=dev), 0)
    eps = ops.logistic_from_uniform(torch.rand(B * N, D, generator=g, device=dev))
    gz, gl = torch.randn(B, N, D, generator=g, device=dev), torch.randn(B, generator=g, device=dev)
    ws = torch.zeros(int(lib.cnf_encoder_bwd_tiled_workspace_floats(B, N, D, C)), device=dev)
    out = torch.empty_like(table)
    cpl = ops.encoder_forward(categ, eps, table, prior, want_class_prob=True)[2]
    lib.cnf_set_encoder_bwd_kernel(variant)
    for _ in range(3):
        _launch(dev, "cnf_encoder_forward_bwd_cpl", _ptr(categ), _ptr(eps), _ptr(table), _ptr(prior), None, 1.0, _ptr(cpl), _ptr(gz), _ptr(gl),
                _ptr(out), _ptr(ws), B, N, D, C, float(ops.LOGISTIC_SIGMA), float(ops.LOGISTIC_LOG_SIGMA), _stream(dev))
    torch.cuda.synchronize()
    lib.cnf_set_encoder_bwd_kernel(0)
    # number of workgroups: the probe region starts behind wgs * C * 2D floats; find it as the first position (multiple of C*2D)
    # whose 8-byte words look like tick counts — simpler: try every candidate count and take the one with plausible sums
    P = C * 2 * D
    best = None
    for wgs in range(1, 2049):
        off = wgs * P
        if off % 2:
            continue
        n = wgs * (NT // 64) * 10
        if off + 2 * n > ws.numel():
            break
        v = ws[off:off + 2 * n].view(torch.int64).view(wgs, NT // 64, 10)
        if int(v.min()) >= 0 and int(v.max()) < 10 ** 7 and int((v.sum(-1) > 0).all()):
            tail = ws[off + 2 * n:off + 2 * n + 16]
            if float(tail.abs().max()) == 0.0:
                best = (wgs, v.double().cpu())
    if best is None:
        print("no probe data (is this a -DCNF_ENC_BWD_PROBE build?)")
        continue
    wgs, v = best
    v = v / 100.0          # 100 MHz ticks -> us
    print("B=%d N=%d D=%d C=%d variant %d: %d workgroups x %d waves; us per wave summed over its stages (mean over workgroups | max)" % (B, N, D, C, variant, wgs, NT // 64))
    print("   %-22s %s" % ("phase", "  ".join("wave %d        " % w for w in range(NT // 64))))
    for i, name in enumerate(NAMES):
        print("   %-22s %s" % (name, "  ".join("%6.1f | %6.1f" % (float(v[:, w, i].mean()), float(v[:, w, i].max())) for w in range(NT // 64))))
    tot = v.sum(-1)
    print("   %-22s %s" % ("total", "  ".join("%6.1f | %6.1f" % (float(tot[:, w].mean()), float(tot[:, w].max())) for w in range(NT // 64))))
