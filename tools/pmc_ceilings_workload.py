"""Workload for the "which ceiling binds" table: each hot kernel at its benchmark shape, REP
launches on rotating buffers, so that rocprofv3 can count VALU / transcendental instructions, busy cycles and HBM
bytes per launch.  Writes the manifest (kernel-name fragment -> algorithmic bytes, elements) next to the counters."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import ops
from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer

dev = torch.device("cuda:0")
REP, R = 12, 4
g = torch.Generator(device=dev).manual_seed(0)
manifest = {}


def run(launches):
    for i in range(REP):
        launches[i % len(launches)]()
    torch.cuda.synchronize()


# affine coupling at S* (the bench's kernels)
B, N, D = 16384, 64, 6
zs = [torch.randn(B, N, D, generator=g, device=dev) for _ in range(R)]
nns = [0.5 * torch.randn(B, N, 2 * D, generator=g, device=dev) for _ in range(R)]
sf, mask = torch.zeros(D, device=dev), CouplingLayer.create_channel_mask(D).to(dev)
zo, lo = torch.empty_like(zs[0]), torch.empty(B, device=dev)
ln = torch.full((B,), float(N), device=dev)
neglog, nll = torch.empty(B, device=dev), torch.empty(B, device=dev)
acc = torch.zeros(ops.NLL_ACC_SLOTS, dtype=torch.int64, device=dev)
run([ops.affine_coupling_nll_acc_launch(zs[r], nns[r], sf, mask, zo, lo, ln, neglog, nll, acc) for r in range(R)])
run([ops.affine_coupling_launch(zs[r], nns[r], sf, mask, zo, lo, reverse=True) for r in range(R)])
e = B * N * D
manifest["affine_coupling_kernel<4, 2, true, false, true, 1>"] = {"what": "affine fwd + NLL + batch sum, S*", "alg_bytes": 16 * e + 4 * B, "elems": e}
manifest["affine_coupling_kernel<4, 2, true, true, true, 0>"] = {"what": "affine inverse, S*", "alg_bytes": 16 * e + 4 * B, "elems": e}

# encoder forward / decode at S*, C = 16
C = 16
cats = [torch.randint(0, C, (B, N), generator=g, device=dev) for _ in range(R)]
table = torch.randn(C, 2 * D, generator=g, device=dev)
prior = torch.log_softmax(torch.zeros(C, device=dev), 0)
for i in range(REP):
    ops.encoder_forward(cats[i % R], zs[i % R], table, prior)
for i in range(REP):
    ops.encoder_decode(zs[i % R], table, prior)
torch.cuda.synchronize()
manifest["encoder_forward_kernel<6>"] = {"what": "encoder forward C=16, S*", "alg_bytes": (8 + 8 * D) * B * N, "elems": B * N, "unit": "tokens"}
manifest["encoder_decode_kernel<6>"] = {"what": "encoder decode C=16, S*", "alg_bytes": (8 + 4 * D) * B * N, "elems": B * N, "unit": "tokens"}
del zs, nns, cats

# mixture coupling: configs[1] and the PTB shape
for tag, (B, N, D, K, masked) in {"configs1": (16384, 16, 4, 8, True), "ptb": (128, 288, 3, 51, False)}.items():
    zs = [torch.randn(B, N, D, generator=g, device=dev) for _ in range(R)]
    nns = [0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=g, device=dev) for _ in range(R)]
    mask = CouplingLayer.create_channel_mask(D).to(dev) if masked else None
    zo, zr = torch.empty_like(zs[0]), torch.empty_like(zs[0])
    lf, lr = torch.empty(B, device=dev), torch.empty(B, device=dev)
    run([ops.mixture_coupling_launch(zs[r], nns[r], mask, K, zo, lf) for r in range(R)])
    run([ops.mixture_coupling_launch(zo, nns[r], mask, K, zr, lr, reverse=True) for r in range(R)])
    e = B * N * D
    # K = 8: exact instantiation; K = 51: 13 predicated slots on each of 4 lanes (cnf_mixture_tok.hip, slots_for)
    kt, gl, pr = (K, 1, "false") if K in (4, 8, 16) else (13, 4, "true")
    manifest["mixture_tok_kernel<%d, false, %d, false, 0, %s>" % (kt, gl, pr)] = {"what": "mixture fwd %s" % tag, "alg_bytes": (16 + 12 * K) * e, "elems": e}
    manifest["mixture_tok_kernel<%d, true, %d, false, 0, %s>" % (kt, gl, pr)] = {"what": "mixture inverse %s" % tag, "alg_bytes": (16 + 12 * K) * e, "elems": e}
    del zs, nns
out = os.environ.get("CNF_MANIFEST")
if out:
    json.dump(manifest, open(out, "w"), indent=1)
print("done")
