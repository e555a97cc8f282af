"""Running the reference's OWN code next to the MI355X layers.

Two things live here, both about the user's checkout of phlippe/CategoricalNF on `sys.path` (nothing of it is copied
into this package):

1. `reference_module(name)` imports one of the reference's source files under a private module name, with the
   handful of one-token fixes its code needs on torch >= 2 applied to the source text IN MEMORY before it is compiled
   (`PATCHES`: an integer division that became a true division, a long tensor clamped with a float bound).  This is
   how the coupling sub-networks that stay plain PyTorch (Edge-GNN, RGCN, LSTM: dense GEMMs / attention, run by
   PyTorch-ROCm) are taken from the reference instead of being re-typed here, e.g.

       gl = categoricalnf_amd.compat.reference_module("layers.networks.graph_layers")
       model = GraphCNF(params, dataset, edge_subnet=lambda stage, c_out_nodes, c_out_edges: gl.EdgeGNN(...))

2. `fall_through(module_globals, name)` is the module-level `__getattr__` of the drop-in modules whose reference
   counterparts also hold host-side helpers that are none of this package's business (the argparse flag builders of the
   reference's training template, an explicit list of names per module): after `categoricalnf_amd.install()`, `from
   layers.flows.distributions import add_prior_distribution_parameters` still works — it is served by the reference's
   own file when the checkout is on sys.path, and is a clear AttributeError otherwise."""
import importlib.util
import os
import sys
import types

# (old, new) text replacements per reference module; each is the minimal change that restores the torch 1.x behaviour
PATCHES = {
    # a long index divided with `/` is a float tensor since torch 1.5 and cannot index (graph_layers.py:527, :668)
    "layers.networks.graph_layers": [("* edge_indices[...,0]) / 2 +", "* edge_indices[...,0]) // 2 +")],
    # a long tensor clamped with a float bound is promoted to float and then refused by scatter_ (general/mutils.py:300)
    "general.mutils": [("inv_time_range = inv_time_range.clamp(min=0.0)", "inv_time_range = inv_time_range.clamp(min=0)"),
                       # torch >= 2.6 loads with weights_only=True by default and refuses the numpy scalars the template
                       # stores in `best_save_dict` / `evaluation_dict` (load_model, general/mutils.py:80, :82)
                       ("torch.load(checkpoint_file)", "torch.load(checkpoint_file, weights_only=False)"),
                       ("torch.load(checkpoint_file, map_location='cpu')",
                        "torch.load(checkpoint_file, map_location='cpu', weights_only=False)")],
    # torch >= 2.2: data.Sampler.__init__ no longer takes the data source (datasets/mutils.py:12, set_summation.py:64)
    "experiments.graph_coloring.datasets.mutils": [("super().__init__(dataset)", "super().__init__()")],
    "experiments.set_modeling.datasets.set_summation": [("super().__init__(dataset)", "super().__init__()")],
    # a float64 numpy prior handed to the last bias (graphCNF.py:77 -> decoder.py:51-54 -> help_layers.py:106-107): torch >= 2
    # refuses to add a double bias to float activations.  Only reached without install(): the drop-in's own
    # layers.networks.help_layers keeps the parameter's dtype
    "layers.networks.help_layers": [("self.main_net[-1].bias.data = bias",
                                     "self.main_net[-1].bias.data = bias.to(self.main_net[-1].bias.dtype)")],
}
_loaded = {}


def find_reference_file(name):
    """Path of the reference source file for dotted module `name` on sys.path (never a file of this package)."""
    rel = os.path.join(*name.split(".")) + ".py"
    here = os.path.dirname(os.path.abspath(__file__))
    for base in sys.path:
        path = os.path.join(base or os.getcwd(), rel)
        if os.path.isfile(path) and not os.path.abspath(path).startswith(here):
            return path
    return None


def apply_patches(name, source, path="<reference>"):
    """The torch >= 2 fixes of `name` applied to its source text.  A fix whose target line is missing AND whose result
    is not there either means another revision of the reference: stop with the file and the line looked for, instead of
    an obscure runtime error inside the reference's code later (used by reference_module and by run_reference's loader)."""
    for old, new in PATCHES.get(name, []):
        if old in source:
            source = source.replace(old, new)
        elif new not in source:
            raise ImportError("%s: cannot apply the torch >= 2 fix for %s — neither %r nor its replacement is in the file "
                              "(another revision of the reference?)" % (path, name, old))
    return source


def reference_module(name):
    """Import the reference's `name` from its checkout on sys.path as `_cnf_reference.<name>`, torch >= 2 fixes applied."""
    if name in _loaded:
        return _loaded[name]
    path = find_reference_file(name)
    if path is None:
        raise ImportError("the reference's %s.py is not on sys.path: add the checkout of phlippe/CategoricalNF to sys.path"
                          % name.replace(".", "/"))
    source = apply_patches(name, open(path).read(), path)
    module = types.ModuleType("_cnf_reference." + name)
    module.__file__ = path
    sys.modules[module.__name__] = module
    _loaded[name] = module
    try:
        exec(compile(source, path, "exec"), module.__dict__)
    except Exception:
        _loaded.pop(name, None)
        sys.modules.pop(module.__name__, None)
        raise
    return module


def fall_through(reference_name, attr):
    """Serve `attr` from the reference's own module `reference_name` (module-level __getattr__ of a drop-in module)."""
    if attr.startswith("__"):
        raise AttributeError(attr)
    try:
        return getattr(reference_module(reference_name), attr)
    except ImportError as e:
        raise AttributeError("%s is host-side code of the reference (%s) that this package does not re-implement, and the "
                             "reference checkout is not on sys.path: %s" % (attr, reference_name, e))
