import sys, traceback
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tests.test_gpu_fuzz as F
fails = 0
for name, fn, gen, extra in [
    ("tiled", F.test_fuzz_encoder_class_tiled, lambda s: F._shapes(s, 40, max_b=40, max_n=24, dims=(1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16)), [()]),
    ("encoder", F.test_fuzz_encoder, lambda s: F._shapes(s, 40, max_b=50, max_n=30, dims=(1,2,3,4,6,8)), [()]),
    ("affine", F.test_fuzz_affine, lambda s: F._shapes(s, 40), [(1,), (0,)]),
    ("mixture", F.test_fuzz_mixture, lambda s: F._shapes(s, 25, max_b=40, max_n=30), [(1,), (0,)]),
    ("linear", F.test_fuzz_actnorm_invconv_prior, lambda s: F._shapes(s, 40), [()]),
]:
    n = 0
    for seed in range(100, 104):
        for shp in gen(seed):
            for ex in extra:
                n += 1
                try:
                    fn(*shp, *ex)
                except Exception as e:
                    fails += 1
                    print("FAIL", name, shp, ex, repr(e)[:300], flush=True)
    print(name, "cases", n, flush=True)
print("total failures", fails)
