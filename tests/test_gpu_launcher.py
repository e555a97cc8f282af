"""GPU test of `categoricalnf_amd.run_reference`, the launcher that runs a script of the user's checkout unchanged.

The reference itself cannot be on the GPU box, so the "checkout" here is a skeleton the test writes into a temporary
directory: the directory layout the launcher looks for, a `general/mutils.py` holding the three lines its torch >= 2
fixes apply to, and an experiment script written the way the reference's are (relative sys.path entry, imports by the
reference's module paths, a tensorboard writer).  What it proves on the device: a script started through the launcher
gets the HIP-backed layers under the reference's import paths and their results equal the CPU oracle's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys
sys.path.append("../../")
import json
import torch
import torch.nn as nn
from torch.utils.tensorboard import SummaryWriter
from general.mutils import bounded
from layers.flows.coupling_layer import CouplingLayer
from layers.flows.activation_normalization import ActNormFlow
import layers.flows.coupling_layer as module

torch.manual_seed(int(sys.argv[sys.argv.index("--seed") + 1]))
D = 6
mask = CouplingLayer.create_channel_mask(D, ratio=0.5)
layer = CouplingLayer(c_in=D, mask=mask, model_func=lambda c_out: nn.Sequential(nn.Linear(D, c_out))).cuda()
with torch.no_grad():
    layer.scaling_factor.normal_(0, 0.3)
z = torch.randn(5, 7, D, device="cuda")
with torch.no_grad():
    out, ldj = layer(z, ldj=None, reverse=False)
    back, ldj_back = layer(out, ldj=None, reverse=True)
    nn_out = layer.nn(z * layer.mask)
writer = SummaryWriter("log")
writer.add_scalar("roundtrip", float((back - z).abs().max()), 0)
writer.close()
torch.save({"z": z.cpu(), "out": out.cpu(), "ldj": ldj.cpu(), "nn_out": nn_out.cpu(), "mask": layer.mask.cpu(),
            "scaling_factor": layer.scaling_factor.detach().cpu(), "ldj_back": ldj_back.cpu()}, "result.pt")
print(json.dumps({"module": module.__name__, "clamped": bounded(torch.tensor([-2, 3])).tolist(),
                  "roundtrip": float((back - z).abs().max()), "argv": sys.argv[1:]}))
'''


def test_launcher_runs_a_checkout_script_on_the_hip_layers(tmp_path):
    import torch
    from categoricalnf_amd import compat
    from oracle import cnf_oracle as O
    root = tmp_path / "checkout"
    for d in ("general", "layers/flows", "experiments/toy"):
        (root / d).mkdir(parents=True)
    (root / "general" / "__init__.py").write_text("")
    (root / "general" / "train.py").write_text("")
    clamp_old = compat.PATCHES["general.mutils"][0][0]
    loads = [o for o, _ in compat.PATCHES["general.mutils"][1:]]
    (root / "general" / "mutils.py").write_text(
        "def bounded(inv_time_range):\n    %s\n    return inv_time_range\nLOADS = \"\"\"%s\"\"\"\n" % (clamp_old, " | ".join(loads)))
    script = root / "experiments" / "toy" / "train.py"
    script.write_text(SCRIPT)
    env = dict(os.environ, PYTHONPATH=ROOT, PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-m", "categoricalnf_amd.run_reference", str(script), "--seed", "3"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=str(tmp_path), timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert "MI355X kernels" in lines[0] and "experiments/toy/train.py" in lines[0]
    rep = json.loads(lines[-1])
    assert rep["module"] == "categoricalnf_amd.layers.flows.coupling_layer"
    assert rep["clamped"] == [0, 3] and rep["argv"] == ["--seed", "3"] and rep["roundtrip"] < 1e-5
    work = root / "experiments" / "toy"                       # the launcher runs the script from its own directory
    scalars = [json.loads(l) for l in open(work / "log" / "scalars.jsonl")]
    assert [s["tag"] for s in scalars] == ["roundtrip"]
    r = torch.load(work / "result.pt")
    z_ref, ldj_ref = O.affine_coupling(r["z"], r["nn_out"], r["mask"], r["scaling_factor"], reverse=False)
    torch.testing.assert_close(r["out"], z_ref, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(r["ldj"], ldj_ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(r["ldj_back"], -ldj_ref, rtol=1e-4, atol=1e-4)
