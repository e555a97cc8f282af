"""categoricalnf_amd — MI355X (gfx950) kernels for the coupling-layer hot path of Categorical
Normalizing Flows, behind the reference's own `layers.flows.*` / `layers.categorical_encoding.*`
module interface.

    import categoricalnf_amd
    categoricalnf_amd.install()          # aliases `layers.flows.*` and `layers.categorical_encoding.*`
    from layers.flows.coupling_layer import CouplingLayer   # -> the HIP-backed module

See INTEGRATION.md.  Importing the package does not need a GPU; running any layer does (no CPU path).
"""
import importlib
import sys

__version__ = "0.1.0"

_FLOW_MODULES = ["flow_layer", "flow_model", "coupling_layer", "mixture_cdf_layer", "autoregressive_coupling",
                 "activation_normalization", "permutation_layers", "distributions", "sigmoid_layer"]
_ENC_MODULES = ["decoder", "linear_encoding", "variational_dequantization", "mutils"]


def install(force=False):
    """Register this package's modules under the reference's import paths.

    `layers.flows`, `layers.categorical_encoding`, `layers.networks.help_layers` and
    `layers.networks.autoregressive_layers` (the LSTM sub-network, which needs the torch >= 2 index-dtype fix of
    host_utils.create_T_one_hot to run at all) resolve to categoricalnf_amd; everything else of the reference
    (`layers.networks.graph_layers`, `general`, `experiments`) is left alone and keeps importing from the reference
    checkout on sys.path."""
    base = __name__ + ".layers"
    pairs = [("layers.flows", base + ".flows"), ("layers.categorical_encoding", base + ".categorical_encoding"),
             ("layers.networks.help_layers", base + ".networks.help_layers"),
             ("layers.networks.autoregressive_layers", base + ".networks.autoregressive_layers")]
    pairs += [("layers.flows." + m, base + ".flows." + m) for m in _FLOW_MODULES]
    pairs += [("layers.categorical_encoding." + m, base + ".categorical_encoding." + m) for m in _ENC_MODULES]
    for alias, real in pairs:
        if alias in sys.modules and not force and sys.modules[alias].__name__ != real:
            raise RuntimeError("%s is already imported from %s; call categoricalnf_amd.install() before importing the "
                               "reference's layers (or pass force=True)" % (alias, sys.modules[alias].__name__))
        sys.modules[alias] = importlib.import_module(real)
    # sub-modules this package does not provide (e.g. layers.categorical_encoding.variational_encoding, which the
    # reference's own mutils.py imports) keep resolving to the reference checkout on sys.path
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    for alias in ("layers.flows", "layers.categorical_encoding"):
        pkg = sys.modules[alias]
        rel = os.path.join(*alias.split("."))
        for entry in sys.path:
            cand = os.path.join(entry or os.getcwd(), rel)
            if os.path.isdir(cand) and not os.path.abspath(cand).startswith(here) and cand not in pkg.__path__:
                pkg.__path__.append(cand)
    return [a for a, _ in pairs]
