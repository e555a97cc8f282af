"""ctypes binding of libcnf_hip.so (C ABI: include/cnf_hip.h).

There is NO fallback: if the library is missing or was not built for this tree the import of any
compute entry point raises.  The library is built in-tree by ``categoricalnf_amd.build.build()``
(``make -C categoricalnf_amd/csrc``); it is git-ignored but travels with the working tree.
"""
import ctypes
import os
import re

import torch  # noqa: F401  (loads PyTorch-ROCm's libamdhip64 first so the kernels share its HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcnf_hip.so")

CNF_OK = 0
CNF_ERR_UNSUPPORTED = 3
FLAG_NAN_Z, FLAG_NAN_LDJ, FLAG_RANGE, FLAG_CATEGORY = 1, 2, 4, 8

_p = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_d = ctypes.c_double
_i64 = ctypes.c_int64

# name -> argtypes, exactly the prototypes of include/cnf_hip.h
SIGNATURES = {
    "cnf_affine_coupling": [_p, _p, _p, _p, _i, _i, _p, _p, _p, _i, _i, _i, _i, _p, _p],
    "cnf_affine_params": [_p, _p, _p, _i, _i, _p, _p, _i, _i, _i, _p],
    "cnf_affine_coupling_actconv": [_p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p],
    "cnf_affine_transform": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p],
    "cnf_actnorm": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p],
    "cnf_ext_actnorm": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p],
    "cnf_actnorm_stats": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "cnf_invconv": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p],
    "cnf_actnorm_invconv": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p],
    "cnf_mixture_coupling": [_p, _p, _p, _p, _p, _i, _i, _p, _i, _p, _i, _i, _p, _p, _p, _p,
                             _i, _i, _i, _i, _i, _d, _d, _i, _p, _p],
    "cnf_mixture_coupling_ws": [_p, _p, _p, _p, _p, _i, _i, _p, _i, _p, _i, _i, _p, _p, _p, _p,
                                _i, _i, _i, _i, _i, _d, _d, _i, _p, _i64, _p, _p],
    "cnf_mixture_coupling_nll": [_p, _p, _p, _p, _p, _i, _i, _p, _i, _p, _i, _i, _p, _p, _p, _p,
                                 _p, _p, _p, _p, _i, _i, _i, _i, _d, _d, _i, _f, _f, _p, _i64, _p, _p],
    "cnf_mixture_coupling_actconv": [_p, _p, _p, _p, _p, _i, _i, _p, _i, _p, _p, _p, _p, _p,
                                     _p, _p, _p, _p, _p, _i, _i, _i, _i, _d, _d, _i, _p, _i64, _p, _p],
    "cnf_mixture_coupling_compact": [_p, _p, _p, _p, _p, _i, _i, _p, _i, _p, _i, _i, _p, _p, _p, _p,
                                     _i, _i, _i, _i, _i, _d, _d, _i, _p, _i64, _p, _p],
    "cnf_mixture_coupling_compact_nll": [_p, _p, _p, _p, _p, _i, _i, _p, _i, _p, _i, _i, _p, _p, _p, _p,
                                         _p, _p, _p, _p, _i, _i, _i, _i, _d, _d, _i, _f, _f, _p, _i64, _p, _p],
    "cnf_mixture_coupling_compact_actconv": [_p, _p, _p, _p, _p, _i, _i, _p, _i, _p, _p, _p, _p, _p,
                                             _p, _p, _p, _p, _p, _i, _i, _i, _i, _d, _d, _i, _p, _i64, _p, _p],
    "cnf_mixture_params": [_p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "cnf_mixture_transform": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _i, _p, _p, _p, _p,
                              _i, _i, _i, _i, _i, _d, _d, _i, _p, _p],
    "cnf_logistic_log_prob": [_p, _p, _i64, _f, _f, _f, _p, _p],
    "cnf_logistic_from_uniform": [_p, _p, _i64, _f, _f, _f, _p],
    "cnf_affine_coupling_nll": [_p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _f, _p, _p],
    "cnf_nll_sum": [_p, _i, _p, _p],
    "cnf_affine_coupling_nll_acc": [_p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _f, _p, _p],
    "cnf_nll_acc_read": [_p, _i64, _d, _p, _p],
    "cnf_prior_nll": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _f, _p],
    "cnf_encoder_forward": [_p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p, _p],
    "cnf_encoder_forward_sampled": [_p, _p, _f, _p, _p, _p, _f, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p, _p],
    "cnf_encoder_decode_actconv": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p, _p],
    "cnf_encoder_forward_actconv": [_p, _p, _f, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p, _p],
    "cnf_encoder_forward_actconv_cpl": [_p, _p, _f, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p, _p],
    "cnf_encoder_decode": [_p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p],
    "cnf_encoder_forward_tiled": [_p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p, _p],
    "cnf_encoder_forward_tiled_sampled": [_p, _p, _f, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p, _p],
    "cnf_encoder_decode_tiled": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p],
    "cnf_sigmoid_flow": [_p, _p, _p, _p, _i, _i, _i, _f, _p, _p],
    "cnf_affine_coupling_bwd": [_p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "cnf_ext_actnorm_bwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "cnf_actnorm_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "cnf_invconv_lu_weight": [_p, _p, _p, _p, _p, _p, _p, _i, _p],
    "cnf_invconv_lu_weight_inv": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _p],
    "cnf_invconv_lu_weight_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p],
    "cnf_actnorm_invconv_bwd": [_p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "cnf_invconv_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "cnf_logistic_log_prob_bwd": [_p, _p, _p, _i64, _f, _f, _p],
    "cnf_prior_nll_bwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p],
    "cnf_sigmoid_flow_bwd": [_p, _p, _p, _p, _i, _i, _i, _f, _p],
    "cnf_mixture_coupling_bwd": [_p, _p, _p, _p, _p, _i, _i, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p,
                                 _i, _i, _i, _i, _d, _d, _i, _p],
    "cnf_mixture_coupling_bwd_f32": [_p, _p, _p, _p, _p, _i, _i, _p, _i, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p,
                                     _i, _i, _i, _i, _d, _d, _i, _p],
    "cnf_mixture_coupling_compact_bwd_f32": [_p, _p, _p, _p, _p, _i, _i, _p, _i, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p,
                                             _i, _i, _i, _i, _d, _d, _i, _p],
    "cnf_bwd_defer_flush": [_p],
    "cnf_encoder_forward_bwd_tiled": [_p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p],
    "cnf_encoder_forward_bwd_cpl": [_p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p],
    "cnf_affine_params_bwd": [_p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "cnf_affine_transform_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "cnf_mixture_transform_bwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                  _i, _i, _i, _i, _d, _d, _i, _p],
    "cnf_mixture_params_bwd": [_p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "cnf_stream_probe": [_p, _p, _p, ctypes.c_long, _i, _p],
    "cnf_stream_probe_bwd": [_p, _p, _p, _p, _p, ctypes.c_long, _i, _i, _p],
    "cnf_probe_affine_fwd_tile": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "cnf_probe_f64_math": [_i, _p, _p, ctypes.c_long, _i, _p],
}
_PLAIN = {"cnf_abi_version": ([], _i), "cnf_last_error": ([], ctypes.c_char_p),
          "cnf_set_tile_chunks": ([_i], None), "cnf_set_unroll": ([_i], None),
          "cnf_set_math_mode": ([_i], None), "cnf_set_inverse_mode": ([_i], None), "cnf_set_mixture_tile": ([_i], None),
          "cnf_set_bwd_tile": ([_i, _i], None), "cnf_set_actnorm_bwd_tiles": ([_i], None), "cnf_set_affine_bwd_tiles": ([_i], None), "cnf_set_mixture_bwd_waves": ([_i], None), "cnf_set_mixture_bwd_big_mb": ([_i], None),
          "cnf_bwd_workspace_floats": ([_i], _i64), "cnf_bwd_defer_begin": ([], None),
          "cnf_mixture_workspace_bytes": ([_i], _i64), "cnf_encoder_workspace_floats": ([_i, _i, _i, _i], _i64),
          "cnf_encoder_bwd_tiled_workspace_floats": ([_i, _i, _i, _i], _i64), "cnf_set_mixture_kernel": ([_i], None), "cnf_set_encoder_kernel": ([_i], None), "cnf_set_encoder_bwd_kernel": ([_i], None), "cnf_encoder_pair_launches": ([], _i64),
          "cnf_set_mixture_lanes": ([_i], None), "cnf_set_mixture_split": ([_i], None),
          "cnf_set_mixture_whole_tokens": ([_i], None), "cnf_set_mixture_nt_mb": ([_i], None),
          "cnf_prof_arm": ([_i], _i), "cnf_prof_collect": ([ctypes.POINTER(ctypes.c_float), _i], _i)}

_lib = None


class CnfLibraryError(RuntimeError):
    pass


def load():
    """Load libcnf_hip.so once; raise CnfLibraryError if it is absent or incomplete."""
    global _lib
    global LIB_PATH
    if _lib is not None:
        return _lib
    if os.environ.get("CNF_LIB_OVERRIDE"):          # A/B of alternative builds of the same ABI (tools/build_variant.sh)
        LIB_PATH = os.environ["CNF_LIB_OVERRIDE"]
    if not os.path.exists(LIB_PATH):
        raise CnfLibraryError(
            "HIP extension %s not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C categoricalnf_amd/csrc`; there is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise CnfLibraryError("%s does not export %s (stale build?)" % (LIB_PATH, name))
        fn.argtypes = argtypes
        fn.restype = _i
    for name, (argtypes, restype) in _PLAIN.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if os.environ.get("CNF_LIB_OVERRIDE"):      # an A/B build of older sources may lack a newer tuning knob
                continue
            raise CnfLibraryError("%s does not export %s (stale build?)" % (LIB_PATH, name))
        fn.argtypes = argtypes
        fn.restype = restype
    runtimes = set(re.findall(r"\S*libamdhip64\S*", open("/proc/self/maps").read()))
    if len(runtimes) > 1:
        raise CnfLibraryError("two HIP runtimes are mapped (%s): streams and pointers would not be shared with "
                              "PyTorch-ROCm" % ", ".join(sorted(runtimes)))
    _lib = lib
    return lib


def exported_symbols():
    """Names declared in include/cnf_hip.h and include/cnf_tuning.h (used by the CPU-side ABI test)."""
    return sorted(list(SIGNATURES) + list(_PLAIN))


def check(status, name):
    if status != CNF_OK:
        raise RuntimeError("%s failed with status %d: %s" % (name, status, load().cnf_last_error().decode()))
