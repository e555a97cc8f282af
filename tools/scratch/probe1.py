import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import cnf_oracle as O
from categoricalnf_amd import ops, _lib
import test_gpu_large_terms as T
lib = _lib.load()
g = lambda t: None if t is None else t.cuda()
B, N, D, K = 3, 703, 2, 8
for pos in [(2, 702, 1), (0, 0, 1), (1, 5, 1), (2, 701, 1)]:
    for val in (3e11, 1e10, 1e3, 200.0, 60.):
        z, nn_out, sf, msf, mask = T._case(B, N, D, K)
        z[pos] = val
        zo, lo, _ = O.mixture_coupling(z, nn_out, mask, K, sf, msf)
        for whole in (1, 0):
            lib.cnf_set_mixture_whole_tokens(whole)
            zf, lf, _ = ops.mixture_coupling(g(z), g(nn_out), g(mask), K, scaling_factor=g(sf), mixture_scaling_factor=g(msf))
            print(pos, val, whole, zo[pos].item(), zf[pos].item(), lo[pos[0]].item(), lf[pos[0]].item())
        lib.cnf_set_mixture_kernel(1)
        zf, lf, _ = ops.mixture_coupling(g(z), g(nn_out), g(mask), K, scaling_factor=g(sf), mixture_scaling_factor=g(msf))
        lib.cnf_set_mixture_kernel(0)
        print('   round1 kernel', zf[pos].item(), lf[pos[0]].item())
        lib.cnf_set_math_mode(0)
        zf, lf, _ = ops.mixture_coupling(g(z), g(nn_out), g(mask), K, scaling_factor=g(sf), mixture_scaling_factor=g(msf))
        lib.cnf_set_math_mode(1)
        print('   mode0', zf[pos].item(), lf[pos[0]].item())
