"""Workload for rocprofv3 counter passes on the encoder backward (cnf_encoder_forward_bwd_tiled) at the benchmark shape
(B=16384, N=64, D=6; C = 16 and 51, or argv): REP calls per launch variant (cnf_set_encoder_bwd_kernel; default 1 = the two
passes and 0 = the shipped choice).  python tools/pmc_encoder_bwd_workload.py [C,C,...] [variant,variant,...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import ops, _lib
from categoricalnf_amd.ops import _ptr, _stream, _launch
dev = torch.device("cuda:0")
lib = _lib.load()
B, N, D = 16384, 64, 6
CS = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (16, 51)
VARIANTS = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 0)
REP = 10
for C in CS:
    g = torch.Generator(device=dev).manual_seed(0)
    categ = torch.randint(0, C, (B, N), generator=g, device=dev)
    table = 0.5 * torch.randn(C, 2 * D, generator=g, device=dev)
    prior = torch.log_softmax(torch.randn(C, generator=g, device=dev), 0)
    eps = ops.logistic_from_uniform(torch.rand(B * N, D, generator=g, device=dev))
    gz, gl = torch.randn(B, N, D, generator=g, device=dev), torch.randn(B, generator=g, device=dev)
    ws = torch.empty(int(lib.cnf_encoder_bwd_tiled_workspace_floats(B, N, D, C)), device=dev)
    out = torch.empty_like(table)
    cpl = ops.encoder_forward(categ, eps, table, prior, want_class_prob=True)[2]
    for v in VARIANTS:                  # v >= 10: variant v - 10 through cnf_encoder_forward_bwd_cpl (the forward's class_prob_log)
        lib.cnf_set_encoder_bwd_kernel(v % 10 if v >= 10 else v)
        for _ in range(REP):
            if v >= 10:
                _launch(dev, "cnf_encoder_forward_bwd_cpl", _ptr(categ), _ptr(eps), _ptr(table), _ptr(prior), None, 1.0, _ptr(cpl), _ptr(gz), _ptr(gl),
                        _ptr(out), _ptr(ws), B, N, D, C, float(ops.LOGISTIC_SIGMA), float(ops.LOGISTIC_LOG_SIGMA), _stream(dev))
            else:
                _launch(dev, "cnf_encoder_forward_bwd_tiled", _ptr(categ), _ptr(eps), _ptr(table), _ptr(prior), None, 1.0, _ptr(gz), _ptr(gl),
                        _ptr(out), _ptr(ws), B, N, D, C, float(ops.LOGISTIC_SIGMA), float(ops.LOGISTIC_LOG_SIGMA), _stream(dev))
        torch.cuda.synchronize()
lib.cnf_set_encoder_bwd_kernel(0)
