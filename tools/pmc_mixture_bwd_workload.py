"""Counter workload for tools/ceilings.py: the fp32 mixture-coupling backward (streaming kernel) at S*, both wave modes.   bash tools/pmc_ceilings.sh ceilings_mixbwd python tools/pmc_mixture_bwd_workload.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import _lib, ops
from categoricalnf_amd.ops import _ptr, _stream
from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
dev = torch.device("cuda:0"); lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(0)
manifest = {}
# (one shape per kernel name: tools/ceilings.py averages the launches of a name)
for tag, (B, N, D, K, waves) in {"S* (4 waves/SIMD build, 34 VGPRs spilled)": (16384, 64, 6, 8, 1), "S* (3 waves/SIMD build, no spills)": (16384, 64, 6, 8, 0)}.items():
    e = B * N * D
    z = torch.randn(B, N, D, generator=g, device=dev)
    nn_ = 0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=g, device=dev)
    mask = CouplingLayer.create_channel_mask(D).to(dev)
    m, mr, mc = ops._mask_desc(mask, D, dev)
    act, n_act = ops._act_list(mask, m, mr, mc, D)
    gz, gl = torch.randn(B, N, D, generator=g, device=dev), torch.randn(B, generator=g, device=dev)
    g_z, g_nn = torch.empty_like(z), torch.empty_like(nn_)
    sf0, msf0 = torch.zeros(D, device=dev), torch.zeros(D, K, device=dev)
    g_sf, g_msf = torch.empty_like(sf0), torch.empty_like(msf0)
    ws = torch.empty(int(lib.cnf_bwd_workspace_floats(D + D * K)), device=dev)
    lib.cnf_set_mixture_bwd_waves(waves)
    for _ in range(8):
        rc = lib.cnf_mixture_coupling_bwd_f32(_ptr(z), _ptr(nn_), _ptr(sf0), _ptr(msf0), _ptr(m), mr, mc, act, n_act, None, 0, 0, _ptr(gz), _ptr(gl),
                                              _ptr(g_z), _ptr(g_nn), _ptr(g_sf), _ptr(g_msf), _ptr(ws), B, N, D, K, -1.0, 1.0, 1, _stream(dev))
        assert rc == 0, lib.cnf_last_error()
    torch.cuda.synchronize()
    lib.cnf_set_mixture_bwd_waves(-1)
    name = "mixture_tok_bwd_kernel%s<8, 1>" % ("_w4" if waves else "")
    manifest[name] = {"what": "mixture bwd " + tag, "alg_bytes": e * (16 + 24 * K) + 8 * e, "elems": e}
    del z, nn_, g_nn
if os.environ.get("CNF_MANIFEST"):
    json.dump(manifest, open(os.environ["CNF_MANIFEST"], "w"), indent=1)
print("done")
