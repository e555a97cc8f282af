"""RCCL over real devices (the reference: nn.DataParallel, general/train.py:36-44, general/mutils.py:243-249).
Every test here needs at least two GPUs and SKIPS on the one-GPU boxes this repository is developed on — on the first
multi-GPU box that runs `pytest -m gpu` they are the first evidence that the N-rank paths work with backend "nccl" (= RCCL on
ROCm) across devices, without anyone editing a file.  The same paths run on every box with all ranks on cuda:0 over gloo
(tests/test_gpu_parity.py: --share-device)."""
import json
import os
import re
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _need(n):
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        pytest.skip("needs %d GPUs for RCCL across devices, this box has %d" % (n, have))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC
    env["PYTHONPATH"] = ROOT
    return env


def _torchrun(n, *args):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(_port())] + list(args)


@pytest.mark.parametrize("n", [2, 4, 8])
def test_bench_runs_n_ranks_over_rccl(n):
    """`python bench.py --gpus N` as the driver launches it: N ranks, one per device, backend nccl; one JSON line with N
    per-rank kernel times measured on N different devices and a finite all-reduce latency."""
    _need(n)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--backend", "nccl", "--steps", "20", "--warmup", "5",
                        "--no-cpu-baseline", "--no-mixture", "--no-kernel-table"], capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["scaling"] == "weak"
    k = out["roofline"]["per_rank_kernel_ms"]
    assert len(k) == n and all(v > 0 for v in k) and len(out["per_rank_elems_per_s"]) == n
    assert out["allreduce_latency_us"] is not None and 0 < out["allreduce_latency_us"] < 1e5
    assert out["value"] > 0 and out["mean_nll"] == out["mean_nll"]
    # the same launch under torchrun (the driver's other spelling)
    r = subprocess.run(_torchrun(n, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "10", "--warmup", "3", "--no-cpu-baseline",
                                 "--no-mixture", "--no-kernel-table"), capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["n_gpus"] == n


def test_data_parallel_gradients_over_rccl():
    """tools/ddp_check.py with backend nccl on two devices: DDP's gradient all-reduce around the HIP backward kernels equals the
    whole batch in one process."""
    _need(2)
    r = subprocess.run(_torchrun(2, os.path.join(ROOT, "tools", "ddp_check.py"), "--backend", "nccl"), capture_output=True, text=True,
                       timeout=900, env=_env(), cwd=ROOT)
    assert r.returncode == 0 and "DDP_CHECK OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_data_dependent_init_statistics_over_rccl():
    """tools/init_sync_check.py with backend nccl: ActNorm's data-dependent initialisation meets across two devices."""
    _need(2)
    r = subprocess.run(_torchrun(2, os.path.join(ROOT, "tools", "init_sync_check.py"), "--backend", "nccl"), capture_output=True, text=True,
                       timeout=900, env=dict(_env(), OMP_NUM_THREADS="1"), cwd=ROOT)
    assert r.returncode == 0 and r.stdout.count("INIT_SYNC OK") == 2, (r.stdout[-2000:], r.stderr[-2000:])


def test_language_modelling_driver_two_devices_over_rccl(tmp_path):
    """One of the three host loops end to end on two devices (DDP over RCCL, sharded evaluation with one all-reduce, rank 0
    writes the checkpoint)."""
    _need(2)
    r = subprocess.run(_torchrun(2, "-m", "categoricalnf_amd.experiments.run_language_modeling", "--vocab_size", "9", "--source_alpha", "0.3",
                                 "--max_seq_len", "32", "--batch_size", "64", "--num_val", "250", "--coupling_hidden_size", "64",
                                 "--coupling_hidden_layers", "1", "--coupling_num_mixtures", "9", "--variable_length", "--max_iterations", "60",
                                 "--eval_freq", "30", "--print_freq", "30", "--checkpoint_path", str(tmp_path / "lm2")),
                       capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    finals = re.findall(r"final: validation ([0-9.]+) bits per character", r.stdout)
    assert len(finals) == 1 and 1.9 < float(finals[0]) < 4.0, r.stdout[-1500:]
    assert any(f.endswith(".tar") for f in os.listdir(tmp_path / "lm2"))


def test_scale_sweep_on_real_devices(tmp_path):
    """tools/scale_sweep.sh at N = 1, 2 (and 4, 8 where the box has them) over RCCL: scale.jsonl and the table as JSON
    (scale.json: value, ms per step, weak-scaling efficiency, all-reduce latency per N)."""
    _need(2)
    have = torch.cuda.device_count()
    ns = [n for n in (1, 2, 4, 8) if n <= have]
    out = str(tmp_path / "scale")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale_sweep.sh"), out] + [str(n) for n in ns], capture_output=True, text=True,
                       timeout=3600, env=dict(_env(), BENCH_FLAGS="--steps 50 --warmup 10 --no-cpu-baseline --no-mixture --no-kernel-table"), cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    table = json.load(open(os.path.join(out, "scale.json")))
    assert [row["n_gpus"] for row in table["rows"]] == ns
    assert all(0.0 < row["efficiency"] < 1.5 for row in table["rows"])
