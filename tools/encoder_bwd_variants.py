"""A/B of the encoder backward's routes (cnf_set_encoder_bwd_kernel) at the benchmark shape: us per call (all launches of the
call, CUDA events over REP calls, interleaved rounds) and the largest difference to variant 1.  Variants: 1 = the two passes;
2 = the pair kernel behind the library's own pre-pass (cnf_encoder_forward_bwd_tiled); 12 = the pair kernel with the forward's
class_prob_log (cnf_encoder_forward_bwd_cpl, knob 2); 10 = cnf_encoder_forward_bwd_cpl with the library's choice by shape (knob 0).
python tools/encoder_bwd_variants.py [B,N,D,C ...]   (ENC_BWD_VARIANTS=1,12 selects)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import ops, _lib
from categoricalnf_amd.ops import _ptr, _stream, _launch
dev = torch.device("cuda:0")
lib = _lib.load()
SHAPES = ((16384, 64, 6, 16), (16384, 64, 6, 51), (16384, 64, 6, 3), (4096, 64, 4, 27), (256, 64, 6, 16), (16384, 64, 8, 42), (2048, 64, 6, 200))
if len(sys.argv) > 1:
    SHAPES = tuple(tuple(int(v) for v in a.split(",")) for a in sys.argv[1:])
VARIANTS = tuple(int(v) for v in os.environ.get('ENC_BWD_VARIANTS', '1,2,12,10').split(','))      # 16 / 17 = 6 / 7 through cnf_encoder_forward_bwd_cpl with the forward's class_prob_log
REP, ROUNDS = 20, 5
for B, N, D, C in SHAPES:
    g = torch.Generator(device=dev).manual_seed(1)
    categ = torch.randint(0, C, (B, N), generator=g, device=dev)
    table = 0.5 * torch.randn(C, 2 * D, generator=g, device=dev)
    prior = torch.log_softmax(torch.randn(C, generator=g, device=dev), 0)
    eps = ops.logistic_from_uniform(torch.rand(B * N, D, generator=g, device=dev))
    gz, gl = torch.randn(B, N, D, generator=g, device=dev), torch.randn(B, generator=g, device=dev)
    ws = torch.empty(int(lib.cnf_encoder_bwd_tiled_workspace_floats(B, N, D, C)), device=dev)
    out = {v: torch.empty_like(table) for v in VARIANTS}

    cpl = ops.encoder_forward(categ, eps, table, prior, want_class_prob=True)[2]

    def run(v):
        lib.cnf_set_encoder_bwd_kernel(v % 10 if v >= 10 else v)
        if v >= 10:
            _launch(dev, "cnf_encoder_forward_bwd_cpl", _ptr(categ), _ptr(eps), _ptr(table), _ptr(prior), None, 1.0, _ptr(cpl), _ptr(gz), _ptr(gl),
                    _ptr(out[v]), _ptr(ws), B, N, D, C, float(ops.LOGISTIC_SIGMA), float(ops.LOGISTIC_LOG_SIGMA), _stream(dev))
        else:
            _launch(dev, "cnf_encoder_forward_bwd_tiled", _ptr(categ), _ptr(eps), _ptr(table), _ptr(prior), None, 1.0, _ptr(gz), _ptr(gl),
                    _ptr(out[v]), _ptr(ws), B, N, D, C, float(ops.LOGISTIC_SIGMA), float(ops.LOGISTIC_LOG_SIGMA), _stream(dev))
    best = {v: 1e9 for v in VARIANTS}
    for v in VARIANTS:
        run(v)
    torch.cuda.synchronize()
    for _ in range(ROUNDS):
        for v in VARIANTS:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(REP):
                run(v)
            b.record()
            torch.cuda.synchronize()
            best[v] = min(best[v], a.elapsed_time(b) * 1e3 / REP)
    lib.cnf_set_encoder_bwd_kernel(0)
    ref = out[1].double()
    scale = float(ref.abs().max())
    print("B=%d N=%d D=%d C=%d (%d tokens)   max |g| %.3e" % (B, N, D, C, B * N, scale))
    for v in VARIANTS:
        err = float((out[v].double() - ref).abs().max())
        rel = float(((out[v].double() - ref).abs() / (ref.abs() + 1e-3 * scale)).max())
        print("   variant %d: %8.1f us per call   max abs diff vs variant 1 %.3e   max rel %.2e   finite %s" % (v, best[v], err, rel, bool(torch.isfinite(out[v]).all())))
