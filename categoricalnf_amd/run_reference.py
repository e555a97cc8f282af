"""Run one of the reference's OWN command-line scripts, unchanged, on the MI355X layers.

    cd <checkout of phlippe/CategoricalNF>/experiments/set_modeling
    python -m categoricalnf_amd.run_reference train.py --dataset shuffling --max_iterations 100000 ...
    python -m categoricalnf_amd.run_reference --reference_root /path/to/CategoricalNF \
           experiments/graph_coloring/train.py --dataset tiny_3 ...

Everything after the script name is the reference's own command line (general/train.py:329-361 and the experiment's
flags); the script file, `general/*`, `experiments/*` and `layers/networks/graph_layers.py` are the user's checkout —
nothing of it is copied here.  Before the script starts this launcher

1. puts the checkout's root on `sys.path` and calls `categoricalnf_amd.install()`, so `layers.flows.*` /
   `layers.categorical_encoding.*` resolve to the HIP-backed modules (`--no_install` leaves the reference's own layers
   in place: the same launcher then runs the plain reference, e.g. on the CPU, for an A/B of the two);
2. registers an import hook that applies `compat.PATCHES` — the one-token torch >= 2 fixes of `general/mutils.py:300`
   and `layers/networks/graph_layers.py:527,668` — to the source text in memory when the script imports those modules
   under their real names;
3. stands in for `torch.utils.tensorboard` when tensorboard is not installed (general/train.py:17 imports it
   unconditionally): `SummaryWriter.add_scalar` lines go to `<log_dir>/scalars.jsonl`, every other writer call is
   accepted and dropped;
4. changes into the script's directory (the scripts append "../../" to `sys.path` and open `data/...` relative to it,
   README of the reference: "cd experiments/<task>; python train.py ...") and runs it as `__main__`;
   `--workdir DIR` runs it from DIR instead (a read-only checkout, or `data/` and `checkpoints/` kept elsewhere — the
   checkout's root is on `sys.path` either way).

The reference's checkpoints (`checkpoint_%07d.tar`), `param_config.pik` and `results.txt` are written by its own
template, so they are the reference's by construction.  Multi-GPU: the reference's `--use_multi_gpu` wraps the model in
`nn.DataParallel` (general/train.py:36-44), which is one process driving several devices; this package's scaling path
is one process per GPU with RCCL (`categoricalnf_amd.distributed`, the drivers under `categoricalnf_amd.experiments`)."""
import importlib.abc
import importlib.machinery
import importlib.util
import json
import os
import runpy
import sys
import types

from . import compat


class _PatchedSourceLoader(importlib.abc.Loader):
    """Loader of one reference module whose source text gets `compat.PATCHES[name]` applied before it is compiled."""

    def __init__(self, name, path):
        self.name, self.path = name, path

    def create_module(self, spec):
        return None

    def get_source(self, fullname=None):
        return compat.apply_patches(self.name, open(self.path).read(), self.path)

    def exec_module(self, module):
        exec(compile(self.get_source(), self.path, "exec"), module.__dict__)


class PatchFinder(importlib.abc.MetaPathFinder):
    """sys.meta_path entry serving the modules named in `compat.PATCHES` from the checkout, fixes applied."""

    def find_spec(self, fullname, path=None, target=None):
        if fullname not in compat.PATCHES:
            return None
        file = compat.find_reference_file(fullname)
        if file is None:
            return None
        return importlib.util.spec_from_file_location(fullname, file, loader=_PatchedSourceLoader(fullname, file))


def install_patch_finder():
    if not any(isinstance(f, PatchFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, PatchFinder())


class JsonlSummaryWriter:
    """What general/train.py needs of tensorboard's SummaryWriter when tensorboard is absent: scalars are appended to
    `<log_dir>/scalars.jsonl` as {"tag", "value", "step"}; any other `add_*` / `flush` / `close` call is a no-op."""

    def __init__(self, log_dir=None, *args, **kwargs):
        self.log_dir = log_dir or "runs"
        os.makedirs(self.log_dir, exist_ok=True)
        self._file = open(os.path.join(self.log_dir, "scalars.jsonl"), "a")

    def add_scalar(self, tag, scalar_value, global_step=None, *args, **kwargs):
        try:
            value = float(scalar_value)
        except (TypeError, ValueError):
            return
        self._file.write(json.dumps({"tag": tag, "value": value, "step": None if global_step is None else int(global_step)})
                         + "\n")

    def flush(self):
        self._file.flush()

    def close(self):
        if not self._file.closed:
            self._file.close()

    def __getattr__(self, name):
        if name.startswith("add_"):
            return lambda *args, **kwargs: None
        raise AttributeError(name)


def ensure_tensorboard():
    """True if the real `torch.utils.tensorboard` imports; otherwise registers the stand-in and returns False."""
    try:
        import torch.utils.tensorboard  # noqa: F401
        return True
    except Exception:
        pass
    import torch.utils
    stub = types.ModuleType("torch.utils.tensorboard")
    stub.SummaryWriter = JsonlSummaryWriter
    stub.__doc__ = "categoricalnf_amd.run_reference stand-in: tensorboard is not installed"
    sys.modules["torch.utils.tensorboard"] = stub
    torch.utils.tensorboard = stub
    return False


def find_root(script, reference_root=None):
    """The checkout's root: --reference_root, $CNF_REFERENCE_ROOT, or the first ancestor of the script that holds
    `general/train.py` and `layers/flows`."""
    cands = [reference_root, os.environ.get("CNF_REFERENCE_ROOT")]
    d = os.path.dirname(os.path.abspath(script))
    while True:
        cands.append(d)
        parent = os.path.dirname(d)
        if parent == d:
            break
        d = parent
    for c in cands:
        if c and os.path.isfile(os.path.join(c, "general", "train.py")) and os.path.isdir(os.path.join(c, "layers", "flows")):
            return os.path.abspath(c)
    raise SystemExit("run_reference: no checkout of phlippe/CategoricalNF found for %s (looked for general/train.py and "
                     "layers/flows in --reference_root, $CNF_REFERENCE_ROOT and the script's ancestors)" % script)


def prepare(root, install=True):
    """Steps 1-3 of the module docstring; returns what was done (for the banner and the tests)."""
    if root not in sys.path:
        sys.path.insert(0, root)
    done = {"root": root, "installed": [], "tensorboard": None}
    if install:
        import categoricalnf_amd
        done["installed"] = categoricalnf_amd.install()
    install_patch_finder()
    done["tensorboard"] = "tensorboard" if ensure_tensorboard() else "scalars.jsonl stand-in"
    return done


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    reference_root, install, workdir = None, True, None
    while argv and argv[0].startswith("--"):
        flag = argv.pop(0)
        if flag == "--reference_root":
            reference_root = argv.pop(0)
        elif flag == "--no_install":
            install = False
        elif flag == "--workdir":
            workdir = argv.pop(0)
        else:
            raise SystemExit("run_reference: unknown launcher flag %s (the script's own flags go after the script name)" % flag)
    if not argv:
        raise SystemExit(__doc__)
    script = argv.pop(0)
    if not os.path.isfile(script) and reference_root and os.path.isfile(os.path.join(reference_root, script)):
        script = os.path.join(reference_root, script)
    if not os.path.isfile(script):
        raise SystemExit("run_reference: %s is not a file" % script)
    script = os.path.abspath(script)
    workdir = os.path.abspath(workdir) if workdir is not None else None
    root = find_root(script, reference_root)
    done = prepare(root, install)
    print("[categoricalnf_amd] %s on %s; layers: %s; torch >= 2 fixes: %s; summary writer: %s" % (
        os.path.relpath(script, root), root,
        "MI355X kernels (%d module aliases)" % len(done["installed"]) if install else "the reference's own",
        ", ".join(sorted(compat.PATCHES)), done["tensorboard"]), flush=True)
    if workdir is not None:
        os.makedirs(workdir, exist_ok=True)
    os.chdir(workdir if workdir is not None else os.path.dirname(script))
    sys.argv = [script] + argv
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
