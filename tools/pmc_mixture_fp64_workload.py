"""Workload for the fp64 ceiling of the reference-precision mixture path (math mode 0: mixture_cdf_layer.py:62,95-142,173-178,235-264
compute in fp64): REP forward and REP Newton-inverse launches of cnf_mixture_coupling at configs[1] and S*, math mode 0.
Writes the manifest (kernel-name fragment, shape -> algorithmic / needed bytes, elements) for tools/fp64_ceilings.py."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import ops, _lib
from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
dev = torch.device("cuda:0")
lib = _lib.load()
REP = 6
g = torch.Generator(device=dev).manual_seed(0)
manifest = []
lib.cnf_set_math_mode(0)
for tag, (B, N, D, K) in (("configs[1]", (16384, 16, 4, 8)), ("S*", (16384, 64, 6, 8))):
    z = torch.randn(B, N, D, generator=g, device=dev)
    nn_out = 0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=g, device=dev)
    mask = CouplingLayer.create_channel_mask(D).to(dev)
    zo, zr = torch.empty_like(z), torch.empty_like(z)
    lf, lr = torch.empty(B, device=dev), torch.empty(B, device=dev)
    fwd = ops.mixture_coupling_launch(z, nn_out, mask, K, zo, lf)
    inv = ops.mixture_coupling_launch(zo, nn_out, mask, K, zr, lr, reverse=True)
    fwd(); inv()
    torch.cuda.synchronize()
    for _ in range(REP):
        fwd()
    torch.cuda.synchronize()
    for _ in range(REP):
        inv()
    torch.cuda.synchronize()
    DA = D - D // 2
    manifest.append({"tag": tag, "B": B, "N": N, "D": D, "K": K, "elems_transformed": B * N * DA, "alg_bytes": B * N * D * (16 + 12 * K),
                     "needed_bytes": B * N * (DA * (2 + 3 * K) * 4 + 8 * D) + 4 * B, "rep": REP})
lib.cnf_set_math_mode(1)
json.dump(manifest, open(os.environ.get("CNF_MANIFEST", "/tmp/fp64_manifest.json"), "w"))
