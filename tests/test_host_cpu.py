"""CPU-side tests (no GPU): C-ABI surface, loader, host logic of the drop-in modules, and the
world_size-2 gloo path of the sharded NLL reduction."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

from tests.golden_util import load_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from categoricalnf_amd import build, _lib
    build.build()
    return _lib.load()


def test_abi_exports_every_declared_symbol(built_lib):
    from categoricalnf_amd import _lib
    proto = r"^(?:int64_t|int|void|const char\*)\s+(cnf_\w+)\s*\("
    product = set(re.findall(proto, open(os.path.join(ROOT, "include", "cnf_hip.h")).read(), flags=re.M))
    tuning = set(re.findall(proto, open(os.path.join(ROOT, "include", "cnf_tuning.h")).read(), flags=re.M))
    assert product and tuning and not (product & tuning), "no prototypes found / a name declared twice"
    # the product header holds what a maintainer of the reference binds: no A/B knob, probe or experimental entry point in it
    assert not [n for n in product if re.match(r"cnf_(set_(?!math_mode|inverse_mode)|probe_|stream_probe|prof_|bwd_defer)", n)], sorted(product)
    declared = product | tuning
    assert declared == set(_lib.exported_symbols())
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    exported = set(re.findall(r" T (cnf_\w+)", nm))
    assert declared <= exported, "missing from the .so: %s" % sorted(declared - exported)
    assert built_lib.cnf_abi_version() == 1


def test_abi_rejects_bad_arguments_without_touching_a_gpu(built_lib):
    rc = built_lib.cnf_affine_coupling(None, None, None, None, 0, 0, None, None, None, 4, 4, 4, 0, None, None)
    assert rc == 1 and b"null tensor" in built_lib.cnf_last_error()
    rc = built_lib.cnf_mixture_coupling(None, None, None, None, None, 0, 0, None, 0, None, 0, 0, None, None, None, None,
                                        1, 1, 1, 1, 0, -1.0, 1.0, 1, None, None)
    assert rc == 1


def test_single_hip_runtime_mapped(built_lib):
    maps = open("/proc/self/maps").read()
    assert len(set(re.findall(r"\S*libamdhip64\S*", maps))) == 1


def test_no_cpu_fallback():
    from categoricalnf_amd import ops
    with pytest.raises(ops.HipOnlyError):
        ops.affine_coupling(torch.randn(2, 3, 2), torch.randn(2, 3, 4), None, None)
    with pytest.raises(ops.HipOnlyError):
        ops.logistic_log_prob(torch.randn(5))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "categoricalnf_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert "cnf_oracle" not in src, f


def test_masks_and_helpers_match_reference_semantics():
    from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
    from categoricalnf_amd.host_utils import create_channel_mask, create_transformer_mask, get_param_val, one_hot
    assert CouplingLayer.create_channel_mask(6).tolist() == [[1, 1, 1, 0, 0, 0]]
    assert CouplingLayer.create_channel_mask(3).tolist() == [[1, 0, 0]]
    assert CouplingLayer.create_channel_mask(3, mask_floor=False).tolist() == [[1, 1, 0]]
    assert CouplingLayer.create_chess_mask().tolist() == [[1], [0]]
    for c in load_cases("affine_coupling"):
        kind = c.meta["mask_kind"]
        m = CouplingLayer.create_channel_mask(c.meta["D"]) if kind == "channel" else CouplingLayer.create_chess_mask()
        if c.meta["flip"]:
            m = 1 - m
        assert torch.equal(m, c.mask)
    layer = CouplingLayer(1, CouplingLayer.create_chess_mask(), lambda c_out: nn.Identity())
    assert layer._prepare_mask(layer.mask, torch.zeros(2, 5, 1)).flatten().tolist() == [1, 0, 1, 0, 1]
    ln = torch.tensor([3, 1])
    assert create_channel_mask(ln, 4).shape == (2, 4, 1) and create_channel_mask(ln, 4)[1, :, 0].tolist() == [1, 0, 0, 0]
    assert create_transformer_mask(ln, 4)[0].tolist() == [False, False, False, True]
    assert get_param_val({"a": 1}, "a") == 1 and get_param_val({}, "b", 7, warning_if_default=False) == 7
    assert one_hot(torch.tensor([2, 0]), 3).tolist() == [[0, 0, 1], [1, 0, 0]]


def test_state_dict_names_info_strings_and_factory():
    """Checkpoint compatibility: parameter/buffer names and info() strings equal the reference's."""
    from categoricalnf_amd.layers.flows.flow_model import FlowModel
    from categoricalnf_amd.layers.flows.activation_normalization import ActNormFlow
    from categoricalnf_amd.layers.flows.permutation_layers import InvertibleConv
    from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
    from categoricalnf_amd.layers.flows.mixture_cdf_layer import MixtureCDFCoupling
    from categoricalnf_amd.layers.flows.autoregressive_coupling import AutoregressiveMixtureCDFCoupling
    from categoricalnf_amd.layers.categorical_encoding.mutils import create_encoding
    c = load_cases("flow_stack")[1]
    m = c.meta
    D, hidden = m["D"], m["hidden"]
    mk = lambda c_out: nn.Sequential(nn.Linear(D, hidden), nn.GELU(), nn.Linear(hidden, c_out))
    params = {"use_dequantization": False, "use_variational": False, "num_dimensions": D, "flow_config": {"num_flows": 0}}
    enc = create_encoding(params, dataset_class=None, vocab_size=m["C"])
    assert "use_dequantization" not in params and "use_variational" not in params      # popped, like the reference
    layers = [enc]
    for _ in range(m["flows"]):
        layers += [ActNormFlow(D), InvertibleConv(D), CouplingLayer(D, CouplingLayer.create_channel_mask(D), mk)]
    model = FlowModel(layers)
    ref_keys = sorted(k[3:] for k in c if k.startswith("sd_"))
    assert sorted(model.state_dict().keys()) == ref_keys
    model.load_state_dict({k[3:]: v for k, v in c.items() if k.startswith("sd_")})
    assert [l.info() for l in model.flow_layers] == m["infos"]
    assert model.need_data_init()
    mix = MixtureCDFCoupling(c_in=4, mask=CouplingLayer.create_channel_mask(4), model_func=lambda c_out: nn.Linear(4, c_out),
                             block_type="Transformer", num_mixtures=8)
    assert set(dict(mix.named_parameters())) == {"scaling_factor", "mixture_scaling_factor", "nn.weight", "nn.bias"}
    assert mix.nn.out_features == 4 * (2 + 3 * 8) and tuple(mix.mixture_scaling_factor.shape) == (4, 8)
    assert mix.info() == "Mixture CDF Coupling Layer - Input size 4, block type Transformer, 8 mixtures, mask ratio 0.50, channel mask"
    ar = AutoregressiveMixtureCDFCoupling(c_in=3, model_func=lambda c_out: nn.Linear(3, c_out), num_mixtures=51)
    assert ar.nn.out_features == 3 * 155
    with pytest.raises(NotImplementedError):
        ar(torch.zeros(1, 2, 3), reverse=True)


def test_install_aliases_reference_import_paths():
    code = ("import categoricalnf_amd as c; c.install();"
            "from layers.flows.coupling_layer import CouplingLayer;"
            "from layers.categorical_encoding.mutils import create_encoding;"
            "from layers.flows.mixture_cdf_layer import MixtureCDFCoupling;"
            "print(CouplingLayer.__module__, create_encoding.__module__)")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert out.returncode == 0, out.stderr
    assert "categoricalnf_amd.layers.flows.coupling_layer" in out.stdout


def test_row_tiling_covers_every_row_once():
    """Host logic of the kernels' tiling mirrored in Python: every row is owned by exactly one tile."""
    from categoricalnf_amd.distributed import shard_bounds
    for total, world in [(16384, 8), (10, 3), (7, 8), (1, 2)]:
        cover = []
        for r in range(world):
            lo, hi = shard_bounds(total, r, world)
            cover += list(range(lo, hi))
        assert cover == list(range(total))


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from categoricalnf_amd.distributed import init_process_group, shard_bounds, allreduce_nll
from oracle import cnf_oracle as O          # test infrastructure: per-rank compute on CPU
rank, _, world = init_process_group("gloo")
g = torch.Generator().manual_seed(0)
B, N, D = 64, 8, 4
z, ldj = torch.randn(B, N, D, generator=g), torch.randn(B, generator=g)
ln = torch.full((B,), N)
lo, hi = shard_bounds(B, rank, world)
nll = O.nll_per_sample(z[lo:hi], ldj[lo:hi], ln[lo:hi])
sums = torch.tensor([float(nll.double().sum()), float(hi - lo)], dtype=torch.float64)
mean, bpd = allreduce_nll(sums)
full = O.nll_per_sample(z, ldj, ln).double().mean().item()
assert abs(mean - full) < 1e-9, (mean, full)
assert abs(bpd - O.bits_per_dim(full)) < 1e-9
# opt-in all-reduce of ActNorm's init statistics (distributed.sync_data_init): the two passes of the data-dependent init
# on row shards — (sum x, count), then sum (x - mean)^2 — against the whole batch (activation_normalization.py:55-67)
from categoricalnf_amd.distributed import allreduce_init_stats, sync_data_init
x = z[lo:hi].double().reshape(-1, D)
acc = torch.cat([x.sum(0), torch.tensor([float(x.shape[0])], dtype=torch.float64)])
assert torch.equal(allreduce_init_stats(acc.clone()), acc)          # switched off: untouched
assert sync_data_init(True) is False
allreduce_init_stats(acc)
mean_c = acc[:D] / acc[D]
acc2 = torch.cat([((x - mean_c) ** 2).sum(0), torch.zeros(1, dtype=torch.float64)])
allreduce_init_stats(acc2)
sync_data_init(False)
zz = z.double().reshape(-1, D)
assert acc[D].item() == B * N and torch.allclose(mean_c, zz.mean(0), atol=1e-12)
assert torch.allclose(acc2[:D] / acc[D], zz.var(0, unbiased=False), atol=1e-12)
if rank == 0:
    print("OK", world, mean)
dist.destroy_process_group()
'''


def test_two_rank_gloo_nll_allreduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29631", str(script), ROOT]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "OK 2" in out.stdout


def test_set_modeling_assembly_matches_reference_checkpoint_keys():
    import json
    data = np.load(os.path.join(ROOT, "tests", "golden", "set_shuffling_model.npz"))
    meta = json.loads(bytes(data["meta"]).decode())
    from categoricalnf_amd.experiments.set_modeling import FlowSetModeling, SetShufflingDataset
    params = {"set_size": meta["set_size"], "coupling_hidden_layers": meta["transformer_layers"],
              "coupling_hidden_size": meta["hidden"], "coupling_num_flows": meta["flows"], "coupling_mask_ratio": 0.5,
              "coupling_num_mixtures": meta["K"],
              "categ_encoding": {"use_dequantization": False, "use_variational": False, "use_decoder": False,
                                 "num_dimensions": meta["D"], "flow_config": {"num_flows": 0}, "decoder_config": {}}}
    model = FlowSetModeling(params, SetShufflingDataset)
    sd = {k[3:]: torch.from_numpy(np.array(data[k])) for k in data.files if k.startswith("sd_")}
    assert sorted(model.state_dict().keys()) == sorted(sd.keys())
    model.load_state_dict(sd)
    assert [l.info() for l in model.flow_layers] == meta["infos"]
    val = SetShufflingDataset(meta["set_size"], train=False, val=True).shuffle_set
    assert val.shape == (32768, 16) and (np.sort(val, axis=1) == np.arange(16)).all()
    assert np.array_equal(val[:256], data["x256"])            # same deterministic validation set as the reference
    assert abs(SetShufflingDataset.optimum_bpd(16) - meta["optimum_bpd"]) < 1e-12


@pytest.mark.skipif(not os.path.isdir("/root/reference/experiments"), reason="reference checkout only exists in the build container")
def test_reference_experiment_code_runs_on_the_dropin_layers():
    """The reference's OWN experiments/set_modeling/flow_model.py, unmodified, assembled on these layers."""
    code = r'''
import sys, io, contextlib
sys.path.insert(0, "%s"); sys.path.insert(0, "/root/reference")
import categoricalnf_amd
categoricalnf_amd.install()
with contextlib.redirect_stdout(io.StringIO()):
    from experiments.set_modeling.flow_model import FlowSetModeling
    from experiments.set_modeling.datasets.set_shuffling import SetShufflingDataset
    import layers.flows.mixture_cdf_layer as m
    assert m.__name__.startswith("categoricalnf_amd"), m.__name__
    p = {"set_size": 16, "coupling_hidden_layers": 1, "coupling_hidden_size": 32, "coupling_num_flows": 2,
         "coupling_mask_ratio": 0.5, "coupling_num_mixtures": 8,
         "categ_encoding": {"use_dequantization": False, "use_variational": False, "use_decoder": False, "num_dimensions": 4,
                            "flow_config": {"num_flows": 0}, "decoder_config": {}}}
    model = FlowSetModeling(p, SetShufflingDataset)
kinds = [type(l).__module__ for l in model.flow_layers]
assert all(k.startswith("categoricalnf_amd") for k in kinds), kinds
print("OK", len(model.state_dict()))
''' % ROOT
    env = dict(os.environ, MPLBACKEND="Agg", PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd="/tmp")
    assert out.returncode == 0, out.stderr[-1500:]
    assert "OK" in out.stdout


def _graph_model(meta):
    from categoricalnf_amd.experiments.graph_coloring import GraphNodeFlow

    class Colours:
        @staticmethod
        def num_node_types():
            return 3

    params = {"coupling_num_flows": meta["flows"], "coupling_hidden_size": meta["hidden"], "coupling_hidden_layers": meta["layers"],
              "coupling_num_mixtures": meta["K"], "coupling_mask_ratio": 0.5, "coupling_dropout": 0.0,
              "categ_encoding": {"use_dequantization": False, "use_variational": False, "use_decoder": False, "num_dimensions": meta["D"],
                                 "flow_config": {"num_flows": 0}, "decoder_config": {}}}
    return GraphNodeFlow(params, Colours)


@pytest.mark.parametrize("c", load_cases("graph_node_flow"))
def test_graph_colouring_assembly_and_rgcn_subnet_match_reference(c):
    """GraphNodeFlow mirror: reference checkpoint keys / info strings, and the RGCN-attention sub-network (a plain
    PyTorch module, dense masked attention here vs. neighbour gathering in the reference) gives the reference's output
    (6..10-, 10..20- and 25..50-node graphs)."""
    model = _graph_model(c.meta)
    sd = {k[3:]: v for k, v in c.items() if k.startswith("sd_")}
    assert sorted(model.state_dict().keys()) == sorted(sd.keys())
    model.load_state_dict(sd)
    assert [l.info() for l in model.flow_layers] == c.meta["infos"]
    from categoricalnf_amd.host_utils import create_channel_mask
    pad = create_channel_mask(c.length, max_len=c.meta["N"])
    with torch.no_grad():
        out = model.flow_layers[3].nn(c.sub_in, adjacency=c.adjacency, channel_padding_mask=pad)
    torch.testing.assert_close(out, c.sub_out, rtol=1e-4, atol=1e-5)


def _language_model(meta):
    from categoricalnf_amd.experiments.language_modeling import FlowLanguageModeling

    class Vocab:
        vectors = None

    params = {"max_seq_len": meta["T"], "coupling_hidden_layers": 1, "coupling_hidden_size": meta["hidden"],
              "coupling_num_flows": meta["flows"], "coupling_num_mixtures": meta["K"], "coupling_dropout": 0.0,
              "coupling_input_dropout": 0.0,
              "categ_encoding": {"use_dequantization": False, "use_variational": False, "use_decoder": False,
                                 "num_dimensions": meta["D"],
                                 "flow_config": {"num_flows": meta["enc_flows"], "hidden_layers": 1, "hidden_size": 32},
                                 "decoder_config": {"num_layers": 1, "hidden_size": 64}}}
    return FlowLanguageModeling(params, None, vocab_size=meta["V"], vocab=Vocab())


@pytest.mark.parametrize("c", load_cases("language_model"))
def test_language_model_assembly_and_lstm_subnet_match_reference(c):
    """FlowLanguageModeling mirror: reference checkpoint keys / info strings, and the autoregressive LSTM sub-network
    (plain PyTorch on both sides; masks built by index arithmetic and applied functionally here, the padded LSTM run
    without packing) reproduces the reference's output on variable-length sequences."""
    model = _language_model(c.meta)
    sd = {k[3:]: v for k, v in c.items() if k.startswith("sd_")}
    assert sorted(model.state_dict().keys()) == sorted(sd.keys())
    model.load_state_dict(sd)
    model.eval()
    assert [l.info() for l in model.flow_layers] == c.meta["infos"]
    from categoricalnf_amd.host_utils import create_channel_mask
    pad = create_channel_mask(c.length, max_len=c.meta["T"])
    sub = [l for l in model.flow_layers if l.__class__.__name__ == "AutoregressiveMixtureCDFCoupling"][0].nn
    with torch.no_grad():
        out = sub(x=c.sub_in, length=c.length, channel_padding_mask=pad)
    torch.testing.assert_close(out, c.sub_out, rtol=1e-4, atol=1e-5)


def test_create_T_one_hot_positions_and_distances():
    from categoricalnf_amd.host_utils import create_T_one_hot
    ln = torch.tensor([4, 2])
    oh = create_T_one_hot(ln, dataset_max_len=5)
    assert oh.shape == (2, 4, 10)
    assert oh[0, 1].tolist() == [0, 1, 0, 0, 0, 0, 0, 1, 0, 0]        # position 1, two steps before the end
    assert oh[1, 1].tolist() == [0, 1, 0, 0, 0, 1, 0, 0, 0, 0]        # position 1 = last of a length-2 sequence
    assert oh[1, 2].abs().sum() == 0 and oh[1, 3].abs().sum() == 0   # beyond the end


def test_set_summation_dataset_matches_reference_enumeration():
    """Facts taken from the reference's create_all_examples / calc_optimum (set_summation.py:84-133) for set size 16,
    sum 42: 2200 multisets in its order (SHA-1 of the int64 table), 63 379 974 736 orderings, optimum 2.2427 bpd."""
    import hashlib
    from categoricalnf_amd.experiments.set_modeling import SetSummationDataset, bounded_partitions
    table = np.array(bounded_partitions(42, 16, 16), dtype=np.int64)
    assert table.shape == (2200, 16)
    assert hashlib.sha1(table.tobytes()).hexdigest() == "e808d6c4f3f5654088df25a64ab9ef303eec905e"
    assert len(bounded_partitions(20, 8, 8)) == 58
    val = SetSummationDataset(16, train=False, val=True)
    assert val.num_orderings == 63379974736.0 and abs(val.optimum_bpd() - 2.242706752094922) < 1e-12
    assert val.eval_sets().shape == (32768, 16) and ((val.eval_sets() + 1).sum(axis=1) == 42).all()
    again = SetSummationDataset(16, train=False, val=True)
    assert np.array_equal(val.eval_sets(), again.eval_sets())                  # reproducible
    train = SetSummationDataset(16, train=True)
    b = train.sample(64, np.random.RandomState(0))
    assert b.shape == (64, 16) and ((b + 1).sum(axis=1) == 42).all() and b.min() >= 0 and b.max() <= 15


def test_checkpoint_files_follow_the_reference_format(tmp_path):
    from categoricalnf_amd.experiments.run_set_modeling import checkpoint_file, load_checkpoint, save_checkpoint
    net = torch.nn.Linear(3, 2)
    opt = torch.optim.RAdam(net.parameters(), lr=1e-3)
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 0.99 ** s)
    f = save_checkpoint(str(tmp_path), 1234, net, opt, sch, best_save_dict={"file": None, "metric": 3.0}, evaluation_dict={1000: 3.1})
    assert os.path.basename(f) == "checkpoint_0001234.tar" and f == checkpoint_file(str(tmp_path), 1234)
    blob = torch.load(f, weights_only=False)
    assert set(blob) == {"model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "iteration", "best_save_dict",
                         "evaluation_dict"}
    other = torch.nn.Linear(3, 2)
    extra = load_checkpoint(str(tmp_path), other)                               # directory -> newest file
    assert torch.equal(other.weight, net.weight) and extra["iteration"] == 1234 and extra["evaluation_dict"] == {1000: 3.1}


def test_best_model_and_argument_pickle_round_trip(tmp_path):
    """general/mutils.py:84-90 (`load_best_model`) and :123-133 / train.py:428-432 (`param_config.pik`)."""
    import shutil
    from categoricalnf_amd.experiments import run_set_modeling as R
    run = tmp_path / "run"
    best_net, last_net = torch.nn.Linear(3, 2), torch.nn.Linear(3, 2)
    best = {"file": R.checkpoint_file(str(run), 100), "metric": 2.5}
    R.save_checkpoint(str(run), 100, best_net, best_save_dict=best)
    R.save_checkpoint(str(run), 200, last_net, best_save_dict=best)
    probe = torch.nn.Linear(3, 2)
    assert R.load_checkpoint(str(run), probe)["iteration"] == 200 and torch.equal(probe.weight, last_net.weight)
    assert R.load_checkpoint(str(run), probe, load_best_model=True)["iteration"] == 100 and torch.equal(probe.weight, best_net.weight)
    moved = tmp_path / "moved"                                    # the stored absolute path no longer exists
    shutil.move(str(run), str(moved))
    assert R.load_checkpoint(str(moved), probe, load_best_model=True)["iteration"] == 100
    os.remove(R.checkpoint_file(str(moved), 100))                 # best file gone: warn and use the newest
    assert R.load_checkpoint(str(moved), probe, load_best_model=True)["iteration"] == 200
    args = R.parse(["--dataset", "summation", "--coupling_num_flows", "3"])
    R.save_args(str(moved), args)
    back = R.load_args(R.checkpoint_file(str(moved), 200))
    assert back.dataset == "summation" and back.coupling_num_flows == 3 and vars(back) == vars(args)


def test_every_c_entry_point_is_documented_for_integrators():
    """include/cnf_hip.h and INTEGRATION.md stay in step: every exported function appears in the binding table."""
    import re
    names = set(re.findall(r"\b(cnf_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "cnf_hip.h")).read()
                           + open(os.path.join(ROOT, "include", "cnf_tuning.h")).read()))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = sorted(n for n in names if n not in doc)
    assert not missing, missing


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the refusal on a box without enough GPUs")
def test_bench_refuses_more_gpus_than_visible():
    """`bench.py --gpus N` must either run N ranks or stop: never print an n_gpus=1 line for an N-GPU request."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0
    assert "HIP device(s) visible" in r.stderr and "{" not in r.stdout
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "{" not in r.stdout


def test_traffic_is_reported_only_for_the_sources_it_was_measured_on(tmp_path):
    import json
    import bench
    sha = bench.kernel_sources_sha()
    assert len(sha) == 16
    good, stale = tmp_path / "good.json", tmp_path / "stale.json"
    good.write_text(json.dumps({"kernel_sources_sha": sha, "affine_coupling_fwd_bytes_per_launch": 123.0}))
    stale.write_text(json.dumps({"kernel_sources_sha": "0" * 16, "affine_coupling_fwd_bytes_per_launch": 123.0}))
    assert bench.read_traffic(str(good))[0] == 123.0
    assert bench.read_traffic(str(stale))[0] is None and "other kernel sources" in bench.read_traffic(str(stale))[1]
    assert bench.read_traffic(str(tmp_path / "none.json"))[0] is None


class _InjectNodes(nn.Module):
    def __init__(self):
        super().__init__()
        self.value = None

    def forward(self, x=None, **kwargs):
        return self.value


class _InjectPair(nn.Module):
    def __init__(self):
        super().__init__()
        self.value = None

    def forward(self, **kwargs):
        return self.value


def _graph_cnf_model(c):
    """The molecule GraphCNF of this package with the golden case's weights and injected sub-network outputs."""
    import contextlib, io
    from categoricalnf_amd.experiments.molecule_generation import GraphCNF
    from categoricalnf_amd.experiments.graph_node_edge_coupling import NodeEdgeCoupling
    m = c.meta

    class Molecules:
        max_num_nodes = staticmethod(lambda: m["N"])
        num_node_types = staticmethod(lambda: m["NT"])
        num_edge_types = staticmethod(lambda: m["ET"])
        num_max_neighbours = staticmethod(lambda: m.get("NEIGH", 4))
        get_node_prior = staticmethod(lambda data_root=None: np.array(m.get("node_prior", [0.4, 0.3, 0.15, 0.1, 0.05]), dtype=np.float32))
        get_edge_prior = staticmethod(lambda data_root=None: np.array(m.get("edge_prior", [0.7, 0.2, 0.1]), dtype=np.float32))

    import copy
    params = copy.deepcopy(m["params"])
    for key in ("categ_encoding_nodes", "categ_encoding_edges"):      # create_encoding pops these two from the dict it is given
        params[key].setdefault("use_dequantization", False)
        params[key].setdefault("use_variational", False)
    with contextlib.redirect_stdout(io.StringIO()):
        model = GraphCNF(params, Molecules, node_subnet=lambda c_out: _InjectNodes(),
                         edge_subnet=lambda stage, c_out_nodes, c_out_edges: _InjectPair())
    model.load_state_dict({k[3:]: v for k, v in c.items() if k.startswith("sd_")}, strict=True)
    inj = [c["inj_%02d" % i] for i in range(sum(1 for k in c.keys() if k.startswith("inj_")))]
    it = iter(inj)
    for flows in (model.step1_flows, model.step2_flows, model.step3_flows):
        for layer in flows:
            if isinstance(layer, NodeEdgeCoupling):
                layer.nn.value = (next(it), next(it))
            elif type(layer).__name__ == "MixtureCDFCoupling":
                layer.nn.value = next(it)
    return model.eval()


@pytest.mark.parametrize("fixture", ["graph_cnf", "graph_cnf_zinc"])
def test_graph_cnf_assembly_matches_reference_names_and_edge_list_helpers(fixture):
    """Molecule GraphCNF (configs[4]; the reduced case of round 2 and the real sizes: 38 nodes, 703 pairs, D = 6 / 2,
    K = 16 / 8, 9 node types, flows 4,6,6): the assembly of this package takes the reference's state_dict as is (strict),
    prints the reference's layer descriptions, and its vectorised edge-list helpers agree with the reference's outputs
    (the golden adjacency decodes back through pairs <-> adjacency)."""
    from categoricalnf_amd.experiments.molecule_generation import adjacency2pairs, pairs2adjacency, get_adjacency_indices
    c = load_cases(fixture)[0]
    model = _graph_cnf_model(c)
    infos = [l.info() for l in list(model.step1_flows) + list(model.step2_flows) + list(model.step3_flows)]
    assert infos == c.meta["infos"]
    pairs, (i, j), valid = adjacency2pairs(c.adjacency, c.length)
    N = c.meta["N"]
    ref_i = torch.tensor([a for a in range(N) for b in range(a + 1, N)])
    ref_j = torch.tensor([b for a in range(N) for b in range(a + 1, N)])
    assert torch.equal(i, ref_i) and torch.equal(j, ref_j)
    assert torch.equal(pairs, c.adjacency.reshape(-1, N * N)[:, ref_i + ref_j * N])
    assert torch.equal(valid, ((ref_i[None] < c.length[:, None]) & (ref_j[None] < c.length[:, None])).float())
    assert torch.equal(pairs2adjacency(N, pairs, c.length, (i, j)), c.adjacency)
    assert torch.equal(get_adjacency_indices(N, c.length)[0], valid)
    assert model.edge_virtual_decoder.layers.main_net[-1].bias.dtype == torch.float32


def test_capture_safe_linear_differentiates_like_torch():
    """graphs.capture_safe_linear (the workaround for the hipGraph memset-node fault, profiles/r03_graph_train_root_cause.txt):
    nn.Linear and nn.MultiheadAttention inside it give the gradients PyTorch gives, the bias gradient taken in single-pass
    stages of at most 64 rows (column sums equal to .sum(0) for awkward row counts); F.linear is restored afterwards."""
    import torch.nn.functional as F
    from categoricalnf_amd.graphs import _column_sums, capture_safe_linear
    for M, N in ((1024, 256), (1000, 7), (65536, 3), (997, 5), (64, 3), (1, 4), (4099, 9)):
        m = torch.randn(M, N, dtype=torch.float64, generator=torch.Generator().manual_seed(M))
        assert torch.allclose(_column_sums(m), m.sum(0), atol=1e-9)
    torch.manual_seed(0)
    lin, mha = torch.nn.Linear(8, 5), torch.nn.MultiheadAttention(8, 2, batch_first=True)
    x = torch.randn(3, 70, 8, requires_grad=True)
    plist = [x] + list(lin.parameters()) + list(mha.parameters())

    def grads():
        o, _ = mha(x, x, x)
        return torch.autograd.grad((lin(o) ** 2).sum(), plist)
    ref = grads()
    orig = F.linear
    with capture_safe_linear():
        assert F.linear is not orig
        got = grads()
        with torch.no_grad():
            assert torch.equal(lin(x), orig(x, lin.weight, lin.bias))
    assert F.linear is orig
    for a, b in zip(ref, got):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-5)


def test_gaussian_prior_and_flag_builders_without_a_checkout():
    """create_prior_distribution(GAUSSIAN) works without the reference on sys.path (in-package class whose
    sample() takes the keyword arguments this package's callers pass); the flag builders, which ARE served from the
    checkout, fail with an AttributeError that says so; a misspelled name is a plain AttributeError."""
    from categoricalnf_amd.layers.flows import distributions as D
    from categoricalnf_amd.layers.categorical_encoding import mutils as M
    prior = D.create_prior_distribution({"distribution_type": D.PriorDistribution.GAUSSIAN, "mu": 0.5, "sigma": 2.0})
    assert isinstance(prior, D.GaussianDistribution) and prior.info() == "Gaussian distribution with mu=0.500000 and sigma=2.000000"
    torch.manual_seed(0)
    x, ldj = prior.sample(shape=(4, 3), return_ldj=True, temp=1.0, device=torch.device("cpu"))
    ref = torch.distributions.normal.Normal(0.5, 2.0)
    assert x.shape == (4, 3) and torch.allclose(ldj, -ref.log_prob(x)) and torch.allclose(prior.log_prob(x), ref.log_prob(x))
    saved = list(sys.path)
    sys.path[:] = [p for p in sys.path if "reference" not in p]
    try:
        for mod, name in ((D, "add_prior_distribution_parameters"), (M, "add_encoding_parameters")):
            with pytest.raises(AttributeError, match="reference checkout is not on sys.path"):
                getattr(mod, name)
        with pytest.raises(AttributeError, match="has no attribute"):
            D.GausianDistribution
    finally:
        sys.path[:] = saved


@pytest.mark.skipif(not os.path.isdir("/root/reference/experiments"), reason="reference checkout only exists in the build container")
def test_reference_edge_gnn_runs_on_torch2_through_compat_and_plugs_into_graph_cnf():
    """§8 f-2: the Edge-GNN sub-network is NOT re-typed in this package; categoricalnf_amd.compat imports the reference's
    own graph_layers.py with the integer-division fix applied in memory (its sparse attention stops on torch >= 2
    otherwise) and the molecule GraphCNF of this package takes it as its stage-2/3 sub-network.  Also: the CLI flag builders
    (and only they) fall through to the reference's files after install()."""
    code = r'''
import sys, io, contextlib
sys.path.insert(0, "%s"); sys.path.insert(1, "/root/reference")
import numpy as np, torch
import categoricalnf_amd
from categoricalnf_amd import compat
categoricalnf_amd.install()
from layers.flows.distributions import add_prior_distribution_parameters, GaussianDistribution, LogisticDistribution
from layers.categorical_encoding.mutils import add_encoding_parameters, create_encoding
assert add_prior_distribution_parameters.__module__.startswith("_cnf_reference") and add_encoding_parameters.__module__.startswith("_cnf_reference")
assert GaussianDistribution.__module__.startswith("categoricalnf_amd")          # in the package: needs no checkout (ADVICE r2)
import layers.flows.distributions as dmod
try:
    dmod.no_such_name
    raise SystemExit("a misspelled attribute must not import the reference's file")
except AttributeError as e:
    assert "no attribute" in str(e)
assert LogisticDistribution.__module__.startswith("categoricalnf_amd") and create_encoding.__module__.startswith("categoricalnf_amd")
with contextlib.redirect_stdout(io.StringIO()):
    gl = compat.reference_module("layers.networks.graph_layers")
from categoricalnf_amd.experiments.molecule_generation import GraphCNF, adjacency2pairs
HN, HE = 16, 8


class Molecules:
    max_num_nodes = staticmethod(lambda: 9)
    num_node_types = staticmethod(lambda: 5)
    num_edge_types = staticmethod(lambda: 3)
    num_max_neighbours = staticmethod(lambda: 4)
    get_node_prior = staticmethod(lambda data_root=None: np.full(5, 0.2, dtype=np.float32))
    get_edge_prior = staticmethod(lambda data_root=None: np.full(3, 1 / 3, dtype=np.float32))


def edge_subnet(stage, c_out_nodes, c_out_edges):
    e2n = (lambda: gl.Edge2NodeAttnLayer(hidden_size_nodes=HN, hidden_size_edges=HE, skip_config=2)) if stage == 1 else \
          (lambda: gl.Edge2NodeQKVAttnLayer(hidden_size_nodes=HN, hidden_size_edges=HE, skip_config=2))
    n2e = lambda: gl.Node2EdgePlainLayer(hidden_size_nodes=HN, hidden_size_edges=HE, skip_config=2)
    return gl.EdgeGNN(c_in_nodes=4, c_in_edges=2, c_out_nodes=c_out_nodes, c_out_edges=c_out_edges,
                      edge_gnn_layer_func=lambda: gl.EdgeGNNLayer(edge2node_layer_func=e2n, node2edge_layer_func=n2e),
                      max_neighbours=4, num_layers=1)


enc = lambda d: {"use_dequantization": False, "use_variational": False, "use_decoder": False, "num_dimensions": d,
                 "flow_config": {"num_flows": 0}, "decoder_config": {}}
params = {"categ_encoding_nodes": enc(4), "categ_encoding_edges": enc(2), "encoding_virtual_num_flows": 0, "coupling_hidden_size_nodes": HN,
          "coupling_hidden_size_edges": HE, "coupling_num_flows": "1,1,1", "coupling_hidden_layers": 1, "coupling_num_mixtures_nodes": 4,
          "coupling_num_mixtures_edges": 4}
with contextlib.redirect_stdout(io.StringIO()):
    model = GraphCNF(params, Molecules, edge_subnet=edge_subnet)
B, N = 3, 9
g = torch.Generator().manual_seed(0)
ln = torch.tensor([9, 6, 4])
valid = torch.arange(N)[None] < ln[:, None]
adj = torch.triu((torch.rand(B, N, N, generator=g) < 0.3).long() * torch.randint(1, 4, (B, N, N), generator=g), 1)
adj = (adj + adj.transpose(1, 2)) * (valid[:, None] & valid[:, :, None]).long()
pairs, x_indices, mv = adjacency2pairs(adj, ln)
real = mv * (pairs != 0).float()
coupling = model.step2_flows[2]
zn, ze = torch.randn(B, N, 4, generator=g), torch.randn(B, pairs.size(1), 2, generator=g)
with torch.no_grad():
    out_n, out_e = coupling.nn(z_nodes=zn, z_edges=ze, length=ln, channel_padding_mask=valid.float().unsqueeze(-1), x_indices=x_indices,
                               mask_valid=real, binary_adjacency=(adj > 0).long())
assert out_n.shape == (B, N, coupling.c_out_nodes) and out_e.shape == (B, pairs.size(1), coupling.c_out_edges)
assert torch.isfinite(out_n).all() and torch.isfinite(out_e).all()
print("OK", sum(p.numel() for p in model.parameters()))
''' % ROOT
    env = dict(os.environ, MPLBACKEND="Agg", PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd="/tmp")
    assert out.returncode == 0, out.stderr[-2500:]
    assert "OK" in out.stdout


# ---- graph-colouring host utilities -----------------------------------------------------------
@pytest.fixture
def coloring_golden():
    import random
    from categoricalnf_amd.experiments.graph_coloring_data import GraphColoringDataset as DS
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "graph_coloring_data.npz"))
    saved = (DS.DATASET_NODES, DS.DATASET_ADJACENCIES, DS.DATASET_TRAIN_IDX, DS.DATASET_VAL_IDX, DS.DATASET_TEST_IDX)
    DS.DATASET_NODES, DS.DATASET_ADJACENCIES = gold["nodes"], gold["adjacency"]
    DS.DATASET_TRAIN_IDX, DS.DATASET_VAL_IDX, DS.DATASET_TEST_IDX = gold["train_idx"], gold["val_idx"], gold["test_idx"]
    yield DS, gold, random
    (DS.DATASET_NODES, DS.DATASET_ADJACENCIES, DS.DATASET_TRAIN_IDX, DS.DATASET_VAL_IDX, DS.DATASET_TEST_IDX) = saved


def test_coloring_validity_matches_reference(coloring_golden):
    """graph_coloring.py:114-135 — per-graph validity and the valid ratio of a padded batch, as the reference counts them."""
    from categoricalnf_amd.experiments.graph_coloring_data import coloring_validity
    DS, gold, _ = coloring_golden
    val = DS(val=True)
    items = [val[i] for i in range(len(val))]
    nodes, adj, ln = (np.stack([it[k] for it in items]) for k in range(3))
    assert np.array_equal(nodes, gold["val_nodes"]) and np.array_equal(adj, gold["val_adjacency"])
    assert np.array_equal(ln, gold["val_length"])
    valid = coloring_validity(torch.from_numpy(nodes), torch.from_numpy(adj), torch.from_numpy(ln))
    assert np.array_equal(valid.numpy(), gold["val_valid"])
    assert 0 < valid.sum() < valid.numel()                                   # the set holds both kinds
    ratio = DS.evaluate_generations(torch.from_numpy(nodes), adj, ln)["valid_ratio"]
    assert ratio == float(gold["val_valid_ratio"])
    # without lengths every node counts: a valid graph stays valid when its padding (colour 0, no edges) is included
    full = coloring_validity(nodes, adj)
    assert bool((full.numpy() == gold["val_valid"]).all())


def test_bucket_sampler_visits_graphs_in_the_reference_order(coloring_golden):
    """datasets/mutils.py:9-61 — same index stream under the same np.random seed; batches hold one length bucket."""
    from categoricalnf_amd.experiments.graph_coloring_data import BucketSampler
    DS, gold, _ = coloring_golden
    train = DS(train=True)
    for bs in (16, 7):
        for seed in (0, 5):
            np.random.seed(seed)
            got = np.array(list(iter(BucketSampler(train, bs))), dtype=np.int64)
            assert np.array_equal(got, gold["sampler_bs%d_seed%d" % (bs, seed)])
            assert sorted(got.tolist()) == list(range(len(train)))           # every graph once
    np.random.seed(3)
    batches = np.array(list(train.get_sampler(16, drop_last=True)), dtype=np.int64)
    assert np.array_equal(batches, gold["batch_sampler_bs16_seed3"])
    lengths = (gold["nodes"][gold["train_idx"]] >= 0).sum(-1)
    same = [len(set(lengths[b].tolist())) == 1 for b in batches]
    assert sum(same) >= len(same) - len(set(lengths.tolist()))               # only bucket-boundary batches mix lengths


@pytest.mark.parametrize("order", ["none", "rand", "largest_first", "smallest_first"])
def test_coloring_dataset_items_match_reference(coloring_golden, order):
    """graph_coloring.py:49-84 — colour permutation, padding to 0, node orderings, under the same `random` / np seeds."""
    DS, gold, random = coloring_golden
    ds = DS(train=True, order_graphs=order)
    random.seed(11); np.random.seed(11)
    items = [ds[i] for i in range(24)]
    for k, name in enumerate(("nodes", "adjacency", "length")):
        assert np.array_equal(np.stack([it[k] for it in items]), gold["item_%s_%s" % (order, name)]), name


def test_coloring_dataset_missing_file_message(tmp_path):
    from categoricalnf_amd.experiments.graph_coloring_data import GraphColoringDataset as DS
    saved = DS.DATASET_NODES
    DS.DATASET_NODES = None
    try:
        with pytest.raises(AssertionError, match="could not be loaded due to a missing file"):
            DS(val=True, data_root=str(tmp_path))
    finally:
        DS.DATASET_NODES = saved


def test_planted_colouring_data_set_has_the_reference_file_format(tmp_path):
    """generate_planted_dataset: the two .npz files GraphColoringDataset (and the reference's) read — padded int8 arrays,
    disjoint splits — every stored colouring valid, no isolated node, node counts inside the requested range."""
    from categoricalnf_amd.experiments.graph_coloring_data import GraphColoringDataset as DS, coloring_validity, generate_planted_dataset
    saved = (DS.DATASET_NODES, DS.DATASET_ADJACENCIES, DS.DATASET_TRAIN_IDX, DS.DATASET_VAL_IDX, DS.DATASET_TEST_IDX,
             DS.PREFIX, DS.NUM_COLORS, DS.DATA_FILENAME, DS.IDX_FILENAME)
    try:
        generate_planted_dataset(str(tmp_path), prefix="_tiny", num_colors=3, num_graphs=400, n_min=10, n_max=20, seed=3)
        DS.set_dataset(prefix="_tiny", num_colors=3)
        DS.DATASET_NODES = DS.DATASET_VAL_IDX = None
        arr = np.load(os.path.join(str(tmp_path), DS.DATA_FILENAME))
        assert arr["nodes"].dtype == np.int8 and arr["nodes"].shape == (400, 20) and arr["adjacency"].shape == (400, 20, 20)
        idx = np.load(os.path.join(str(tmp_path), DS.IDX_FILENAME))
        parts = [idx[k] for k in ("train_idx", "val_idx", "test_idx")]
        assert sorted(np.concatenate(parts).tolist()) == list(range(400))
        val = DS(val=True, data_root=str(tmp_path))
        items = [val[i] for i in range(len(val))]
        nodes, adj, ln = (np.stack([it[k] for it in items]) for k in range(3))
        assert ln.min() >= 10 and ln.max() <= 20
        assert bool(coloring_validity(nodes, adj, ln).all())
        inside = np.arange(20)[None, :] < ln[:, None]
        assert bool(((adj > 0).sum(-1) >= 1)[inside].all())                 # no unconstrained node
        assert np.array_equal(adj, adj.transpose(0, 2, 1))
    finally:
        (DS.DATASET_NODES, DS.DATASET_ADJACENCIES, DS.DATASET_TRAIN_IDX, DS.DATASET_VAL_IDX, DS.DATASET_TEST_IDX,
         DS.PREFIX, DS.NUM_COLORS, DS.DATA_FILENAME, DS.IDX_FILENAME) = saved


# ---------------------------------------------------------------- §8 f-3: the reference's own CLI through the launcher

def test_run_reference_summary_writer_and_patch_finder(tmp_path):
    """Launcher pieces that need no checkout: the tensorboard stand-in records scalars and swallows the other writer
    calls; the import hook serves a module named in compat.PATCHES from a checkout with the fix applied, and refuses a
    checkout whose line it cannot find."""
    import importlib
    import json
    from categoricalnf_amd import compat, run_reference
    w = run_reference.JsonlSummaryWriter(str(tmp_path / "log"))
    w.add_scalar("train/loss", torch.tensor(1.5), 7)
    w.add_scalar("train/text", "not a number", 7)
    w.add_histogram("h", torch.zeros(3), 7)
    w.add_text("t", "x")
    w.flush()
    w.close()
    lines = [json.loads(l) for l in open(tmp_path / "log" / "scalars.jsonl")]
    assert lines == [{"tag": "train/loss", "value": 1.5, "step": 7}]
    with pytest.raises(AttributeError):
        w.no_such_call

    root = tmp_path / "checkout"
    (root / "general").mkdir(parents=True)
    (root / "general" / "__init__.py").write_text("")
    old, new = compat.PATCHES["general.mutils"][0]
    loads = [o for o, _ in compat.PATCHES["general.mutils"][1:]]            # the two torch.load call sites
    (root / "general" / "mutils.py").write_text("def f(inv_time_range):\n    %s\n    return inv_time_range\nSRC = \"\"\"%s\"\"\"\n"
                                                % (old, " | ".join(loads)))
    saved_path, saved_meta = list(sys.path), list(sys.meta_path)
    saved_mods = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "general" or k.startswith("general.")}
    try:
        sys.path.insert(0, str(root))
        run_reference.install_patch_finder()
        run_reference.install_patch_finder()                               # idempotent
        assert sum(isinstance(f, run_reference.PatchFinder) for f in sys.meta_path) == 1
        mod = importlib.import_module("general.mutils")
        assert mod.__file__ == str(root / "general" / "mutils.py")
        assert mod.SRC == "torch.load(checkpoint_file, weights_only=False) | torch.load(checkpoint_file, map_location='cpu', weights_only=False)"
        assert mod.f(torch.tensor([-2, 3])).dtype == torch.int64           # clamp(min=0) keeps the long dtype
        (root / "general" / "mutils.py").write_text("x = 1\n")
        sys.modules.pop("general.mutils")
        with pytest.raises(ImportError, match="torch >= 2 fix"):
            importlib.import_module("general.mutils")
    finally:
        sys.path[:], sys.meta_path[:] = saved_path, saved_meta
        for k in [k for k in sys.modules if k == "general" or k.startswith("general.")]:
            sys.modules.pop(k)
        sys.modules.update(saved_mods)


def test_run_reference_finds_the_checkout_root(tmp_path):
    from categoricalnf_amd import run_reference
    root = tmp_path / "ref"
    (root / "general").mkdir(parents=True)
    (root / "general" / "train.py").write_text("")
    (root / "layers" / "flows").mkdir(parents=True)
    script = root / "experiments" / "task" / "train.py"
    script.parent.mkdir(parents=True)
    script.write_text("")
    assert run_reference.find_root(str(script)) == str(root)
    with pytest.raises(SystemExit):
        run_reference.find_root(str(tmp_path / "elsewhere.py"))
    with pytest.raises(SystemExit, match="unknown launcher flag"):
        run_reference.main(["--frobnicate", str(script)])


@pytest.mark.skipif(not os.path.isdir("/root/reference/experiments"), reason="reference checkout only exists in the build container")
def test_reference_cli_runs_unchanged_up_to_the_kernel_boundary(tmp_path):
    """§8 f-3 "run the reference's CLI unchanged": experiments/set_modeling/train.py of the checkout, started through
    the launcher with its own flags, parses them, writes its param_config.pik, builds FlowSetModeling on the drop-in
    layers and the task's data loaders, opens the (stand-in) summary writer and starts its data-dependent
    initialisation — whose first layer call stops at the package's boundary on a machine without a GPU: HipOnlyError,
    no silent CPU path.  (On an MI355X the same command trains; the layers it reaches are covered by the GPU suite.)
    Set summation at set size 16 is BASELINE configs[1]; its `PreSampler` needs the torch >= 2.2 constructor fix."""
    ckpt = tmp_path / "ckpt"
    env = dict(os.environ, MPLBACKEND="Agg", PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=ROOT, CUDA_VISIBLE_DEVICES="",
               HIP_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "-m", "categoricalnf_amd.run_reference", "--reference_root", "/root/reference",
                          "experiments/set_modeling/train.py", "--dataset", "summation",
                          "--max_iterations", "4", "--eval_freq", "2", "--batch_size", "16", "--coupling_num_flows", "2",
                          "--coupling_hidden_size", "32", "--checkpoint_path", str(ckpt), "--cluster"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=str(tmp_path), timeout=600)
    assert out.returncode != 0
    assert "MI355X kernels" in out.stdout and "Initializing data dependent" in out.stdout, out.stdout[-1500:]
    assert "HipOnlyError" in out.stderr and "/root/reference/general/train.py" in out.stderr, out.stderr[-1500:]
    assert "categoricalnf_amd/layers/flows" in out.stderr
    import pickle
    with open(ckpt / "param_config.pik", "rb") as f:
        args = pickle.load(f)                                              # written by general/train.py:428-432
    args = args if isinstance(args, dict) else vars(args)
    assert args["dataset"] == "summation" and args["set_size"] == 16 and args["coupling_num_flows"] == 2
    assert (ckpt / "scalars.jsonl").exists()


@pytest.fixture(scope="module")
def planted_workdir(tmp_path_factory):
    """A working directory outside the (read-only) checkout holding `data/` in the reference's graph-colouring format."""
    from categoricalnf_amd.experiments.graph_coloring_data import generate_planted_dataset
    work = tmp_path_factory.mktemp("gc_work")
    generate_planted_dataset(str(work / "data"), num_graphs=600)
    return work


def _run_graph_colouring_cli(work, ckpt, *launcher_flags):
    env = dict(os.environ, MPLBACKEND="Agg", PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=ROOT, CUDA_VISIBLE_DEVICES="",
               HIP_VISIBLE_DEVICES="")
    return subprocess.run([sys.executable, "-m", "categoricalnf_amd.run_reference", *launcher_flags, "--reference_root",
                           "/root/reference", "--workdir", str(work), "experiments/graph_coloring/train.py", "--dataset", "tiny_3",
                           "--max_iterations", "4", "--eval_freq", "2", "--batch_size", "16", "--coupling_num_flows", "2",
                           "--coupling_hidden_size", "32", "--checkpoint_path", str(ckpt), "--cluster"],
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd="/tmp", timeout=600)


@pytest.mark.skipif(not os.path.isdir("/root/reference/experiments"), reason="reference checkout only exists in the build container")
def test_reference_graph_colouring_cli_trains_through_the_launcher_on_torch2(planted_workdir):
    """The launcher without the drop-in (`--no_install`): the reference's experiments/graph_coloring/train.py trains,
    evaluates, samples, checkpoints, reloads its best checkpoint and tests — on torch 2.10, which it cannot do by itself
    (Sampler constructor, long-index division, torch.load default; compat.PATCHES applied by the import hook) — from a
    working directory outside the checkout, on the data set `generate_planted_dataset` wrote in its file format."""
    ckpt = planted_workdir / "ckpt_plain"
    out = _run_graph_colouring_cli(planted_workdir, ckpt, "--no_install")
    assert out.returncode == 0, out.stderr[-2000:]
    assert "the reference's own" in out.stdout and "Num training examples: 480" in out.stdout
    assert "Validity ratio" in out.stdout and "Reversibility test passed" in out.stdout and "Test performance" in out.stdout
    files = sorted(os.listdir(ckpt))
    assert "param_config.pik" in files and "results.txt" in files and "scalars.jsonl" in files
    assert any(re.fullmatch(r"checkpoint_\d{7}\.tar", f) for f in files)
    import json
    tags = {json.loads(l)["tag"] for l in open(ckpt / "scalars.jsonl")}
    assert any(t.startswith("eval/") for t in tags), sorted(tags)[:10]


@pytest.mark.skipif(not os.path.isdir("/root/reference/experiments"), reason="reference checkout only exists in the build container")
def test_reference_graph_colouring_cli_reaches_the_kernel_boundary(planted_workdir):
    """Same command with the drop-in installed: GraphNodeFlow of the checkout is assembled from this package's layers and
    the checkout's (fixed) RGCN sub-network, the task loads the data, and the first layer call refuses the CPU tensor."""
    out = _run_graph_colouring_cli(planted_workdir, planted_workdir / "ckpt_hip")
    assert out.returncode != 0
    assert "MI355X kernels" in out.stdout and "Preparing data dependent initialization" in out.stdout
    assert "HipOnlyError" in out.stderr and "categoricalnf_amd/layers/" in out.stderr
    assert "/root/reference/experiments/graph_coloring/train.py" in out.stderr


def test_markov_corpus_statistics():
    """The synthetic language-modelling source: stationary pair distribution, entropy rate between 0 and the
    context-free entropy, samples that follow the transition table (cross-entropy of a large sample under the true model
    equals the entropy rate), reproducible under a seed."""
    from categoricalnf_amd.experiments.run_language_modeling import MarkovCorpus
    c = MarkovCorpus(vocab_size=9, alpha=0.3, seed=5)
    assert abs(c.pair_stationary.sum() - 1) < 1e-12
    assert np.abs(np.einsum("ab,abc->bc", c.pair_stationary, c.T) - c.pair_stationary).max() < 1e-12
    assert 0 < c.entropy_rate() < c.unigram_entropy() <= np.log2(9) + 1e-9
    x = c.sample(400, 200, np.random.RandomState(0))
    assert x.shape == (400, 200) and x.min() >= 0 and x.max() < 9
    assert np.array_equal(x, c.sample(400, 200, np.random.RandomState(0)))
    nll = -np.log2(c.T[x[:, :-2], x[:, 1:-1], x[:, 2:]]).mean()
    assert abs(nll - c.entropy_rate()) < 0.03, (nll, c.entropy_rate())


def test_flat_parameters_train_like_per_tensor_parameters():
    """FlatParameters: RAdam + global-norm clipping on one flat buffer == the same on the separate tensors (same updates up
    to the rounding of the clipping norm), gradients accumulate into the views, load_state_dict keeps the views."""
    from categoricalnf_amd.host_utils import FlatParameters

    def make():
        torch.manual_seed(0)
        return nn.Sequential(nn.Linear(5, 16), nn.Tanh(), nn.Linear(16, 16), nn.LayerNorm(16), nn.Linear(16, 3))
    a, b = make(), make()
    flat = FlatParameters(b)
    assert flat.intact() and flat.flat.numel() == sum(p.numel() for p in a.parameters())
    opt_a = torch.optim.RAdam(a.parameters(), lr=1e-2)
    opt_b = torch.optim.RAdam(flat.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(1)
    for _ in range(12):
        x, y = torch.randn(32, 5, generator=g), torch.randn(32, 3, generator=g)
        opt_a.zero_grad(set_to_none=True)
        ((a(x) - y) ** 2).mean().backward()
        na = torch.nn.utils.clip_grad_norm_(a.parameters(), 0.25)
        opt_a.step()
        flat.zero_grad()
        ((b(x) - y) ** 2).mean().backward()
        nb = torch.nn.utils.clip_grad_norm_(flat.parameters(), 0.25)
        opt_b.step()
        assert flat.intact()
        torch.testing.assert_close(na, nb, rtol=1e-5, atol=1e-7)
    for pa, pb in zip(a.parameters(), b.parameters()):
        torch.testing.assert_close(pa, pb, rtol=1e-5, atol=1e-6)
    b.load_state_dict(a.state_dict())
    assert flat.intact()
    torch.testing.assert_close(flat.flat.detach(), torch.cat([p.detach().reshape(-1) for p in a.parameters()]))


def test_molecule_like_dataset_format(tmp_path):
    from categoricalnf_amd.experiments.molecule_data import generate_molecule_like_dataset
    nodes, adj = generate_molecule_like_dataset(str(tmp_path), num_graphs=300, num_val=100, seed=1)
    data = np.load(tmp_path / "zinc250k" / "zinc250k_compressed.npz")
    idx = np.load(tmp_path / "zinc250k" / "zinc250k_dataidx.npz")
    assert np.array_equal(data["nodes"], nodes) and np.array_equal(data["adjacency"], adj)
    assert nodes.shape == (300, 38) and adj.shape == (300, 38, 38) and nodes.dtype == np.int8
    assert sorted(np.concatenate([idx["train_idx"], idx["val_idx"]]).tolist()) == list(range(300))
    length = (nodes >= 0).sum(1)
    assert length.min() >= 8 and length.max() <= 38 and nodes.max() <= 8 and adj.min() == 0 and adj.max() <= 3
    assert (adj == adj.transpose(0, 2, 1)).all() and (adj.sum(-1) > 0).sum(1).tolist() == length.tolist()    # connected, padded rows empty
    assert ((adj > 0).sum(-1).max()) <= 4                                     # at most four bonds per atom


@pytest.mark.skipif(not os.path.isdir("/root/reference/experiments"), reason="reference checkout only exists in the build container")
def test_reference_molecule_cli_reaches_the_kernel_boundary(tmp_path):
    """Third command line through the launcher (configs[4]): the checkout's experiments/molecule_generation/train.py builds
    ITS OWN three-stage GraphCNF (graphCNF.py) from this package's layers and its own Edge-GNN (torch >= 2 fix applied by the
    import hook), reads a data set in the Zinc250k file format, computes its node / edge priors, and starts the
    data-dependent initialisation, whose first kernel call refuses the CPU tensor."""
    from categoricalnf_amd.experiments.molecule_data import generate_molecule_like_dataset
    generate_molecule_like_dataset(str(tmp_path / "data"), num_graphs=9000, seed=0)
    env = dict(os.environ, MPLBACKEND="Agg", PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=ROOT, CUDA_VISIBLE_DEVICES="",
               HIP_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "-m", "categoricalnf_amd.run_reference", "--reference_root", "/root/reference",
                          "--workdir", str(tmp_path), "experiments/molecule_generation/train.py", "--max_iterations", "4",
                          "--eval_freq", "2", "--batch_size", "16", "--coupling_hidden_size_nodes", "32",
                          "--coupling_hidden_size_edges", "16", "--checkpoint_path", str(tmp_path / "ckpt"), "--cluster"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd="/tmp", timeout=600)
    assert out.returncode != 0
    assert "MI355X kernels" in out.stdout and "Preparing data dependent initialization" in out.stdout, out.stdout[-1500:]
    assert "HipOnlyError" in out.stderr and "/root/reference/experiments/molecule_generation/graphCNF.py" in out.stderr
    assert os.path.isfile(tmp_path / "data" / "zinc250k" / "zinc250k_node_prior.npy")      # written by the reference's dataset class


def test_graph_colouring_evaluation_batches_are_dealt_out_once(tmp_path):
    """Sharded evaluation of the graph-colouring driver: the ranks' shares are disjoint, together they are exactly the
    single-process batch list, a graph budget cuts every rank at the same batch, and the caller's numpy random stream is
    untouched (training order does not depend on when evaluations happen)."""
    from categoricalnf_amd.experiments.graph_coloring_data import GraphColoringDataset, generate_planted_dataset
    from categoricalnf_amd.experiments.run_graph_coloring import evaluation_share
    generate_planted_dataset(str(tmp_path), num_graphs=500, seed=3)
    GraphColoringDataset.set_dataset(prefix="_tiny", num_colors=3)
    GraphColoringDataset.DATASET_NODES = GraphColoringDataset.DATASET_VAL_IDX = None
    val = GraphColoringDataset(num_colors=3, val=True, data_root=str(tmp_path))
    np.random.seed(77)
    before = np.random.get_state()[1].copy()
    whole = evaluation_share(val, 16)
    assert np.array_equal(np.random.get_state()[1], before)
    assert sorted(i for b in whole for i in b) == list(range(len(val)))
    for world in (2, 3):
        shares = [evaluation_share(val, 16, r, world) for r in range(world)]
        dealt = [b for k in range(len(whole)) for b in [shares[k % world][k // world]]]
        assert dealt == whole
    budget = [evaluation_share(val, 16, r, 2, max_graphs=20) for r in range(2)]
    assert sum(len(b) for s in budget for b in s) == len(whole[0]) + len(whole[1]) and budget[0][0] == whole[0]
    GraphColoringDataset.DATASET_NODES = GraphColoringDataset.DATASET_VAL_IDX = None


@pytest.mark.skipif(not os.path.isdir("/root/reference/general"), reason="reference checkout only exists in the build container")
def test_driver_beta_schedules_equal_the_reference_scheduler():
    """`beta_at` of the graph-colouring and language-modelling drivers against the reference's ExponentialScheduler object
    (general/parameter_scheduler.py:109-121) built with the defaults of its train.py files (start 1, end 2, step 5000,
    logit 2, no delay)."""
    code = r'''
import sys
sys.path.insert(0, "%s"); sys.path.insert(1, "/root/reference")
from general.parameter_scheduler import ExponentialScheduler
from categoricalnf_amd.experiments import run_graph_coloring as G, run_language_modeling as L
ref = ExponentialScheduler(start_val=1.0, end_val=2.0, logit_factor=2, stepsize=5000, delay=0)
ga, la = G.parse([]), L.parse([])
for it in (0, 1, 17, 2500, 5000, 12345, 99999):
    assert abs(G.beta_at(ga, it) - ref.get(it)) < 1e-12 and abs(L.beta_at(la, it) - ref.get(it)) < 1e-12, it
print("OK")
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd="/tmp",
                         env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-1500:]


def test_language_modelling_batches():
    """draw_batch of the language-modelling driver: fixed-length batches are full; variable-length batches keep their full
    width through the first sentence, lengths stay in [T / 4, T], positions past a length hold the padding symbol, and the
    unpadded prefix is the source's sample."""
    from categoricalnf_amd.experiments import run_language_modeling as L
    corpus = L.MarkovCorpus(vocab_size=7, alpha=0.4, seed=2)
    args = L.parse(["--max_seq_len", "40", "--vocab_size", "7"])
    x, ln = L.draw_batch(corpus, args, 9, np.random.RandomState(4), "cpu")
    assert x.shape == (9, 40) and x.dtype == torch.int64 and ln.tolist() == [40] * 9
    args = L.parse(["--max_seq_len", "40", "--vocab_size", "7", "--variable_length"])
    xv, lv = L.draw_batch(corpus, args, 9, np.random.RandomState(4), "cpu")
    assert lv[0] == 40 and lv.min() >= 10 and lv.max() <= 40 and lv.dtype == torch.int64
    for b in range(9):
        assert (xv[b, lv[b]:] == 0).all() and torch.equal(xv[b, :lv[b]], x[b, :lv[b]])
    assert abs(L.beta_at(args, 5000) - 1.5) < 1e-12 and L.beta_at(args, 0) == 1.0


# ---- the in-package Edge-GNN (sub-network of the molecule flow's edge stages; plain PyTorch, runs on any device) -----------
def _build_edge_gnn(mod, m, c_in_nodes=6, c_in_edges=2):
    hn, he = m["hidden_nodes"], m["hidden_edges"]
    if m["step"] == 1:
        e2n = lambda: mod.Edge2NodeAttnLayer(hidden_size_nodes=hn, hidden_size_edges=he, skip_config=2)          # noqa: E731
    else:
        e2n = lambda: mod.Edge2NodeQKVAttnLayer(hidden_size_nodes=hn, hidden_size_edges=he, skip_config=2)       # noqa: E731
    n2e = lambda: mod.Node2EdgePlainLayer(hidden_size_nodes=hn, hidden_size_edges=he, skip_config=2)             # noqa: E731
    return mod.EdgeGNN(c_in_nodes=c_in_nodes, c_in_edges=c_in_edges, c_out_nodes=m["c_out_nodes"], c_out_edges=m["c_out_edges"],
                       edge_gnn_layer_func=lambda: mod.EdgeGNNLayer(edge2node_layer_func=e2n, node2edge_layer_func=n2e),
                       num_layers=m["layers"], max_neighbours=m["max_neighbours"])


def edge_gnn_from_golden(cases, c, device="cpu"):
    from categoricalnf_amd.layers.networks import edge_gnn
    net = _build_edge_gnn(edge_gnn, c.meta).to(device).eval()
    w = cases[c.meta["weights_case"]]
    net.load_state_dict({k[3:]: v for k, v in w.items() if k.startswith("sd_")}, strict=True)       # the reference's own keys
    return net


def run_edge_gnn_case(net, c, device="cpu"):
    d = lambda t: t.to(device)
    with torch.no_grad():
        return net(d(c.z_nodes), d(c.z_edges), length=d(c.length), x_indices=(d(c.x1), d(c.x2)), mask_valid=d(c.mask_valid),
                   channel_padding_mask=d(c.pad), binary_adjacency=d(c.adjacency) if c.meta["use_adjacency"] else None)


def test_edge_gnn_golden_cpu():
    """layers/networks/edge_gnn.py against the REFERENCE's EdgeGNN outputs at the Zinc250k graph sizes (tests/golden/
    edge_gnn.npz: both node-update layers, with and without binary_adjacency, padded graphs); the reference's state_dict
    loads strictly."""
    from tests.golden_util import load_cases
    cases = load_cases("edge_gnn")
    assert {(c.meta["step"], c.meta["use_adjacency"]) for c in cases} == {(1, True), (1, False), (2, True), (2, False)}
    for c in cases:
        on, oe = run_edge_gnn_case(edge_gnn_from_golden(cases, c), c)
        scale_n, scale_e = c.out_nodes.abs().max().item(), c.out_edges.abs().max().item()
        assert (on - c.out_nodes).abs().max().item() <= 2e-5 * max(scale_n, 1.0), c.meta
        assert (oe - c.out_edges).abs().max().item() <= 2e-5 * max(scale_e, 1.0), c.meta
        # padded nodes and invalid pairs are exactly zero
        assert float((on * (1 - c.pad)).abs().max()) == 0.0 and float((oe * (1 - c.mask_valid.unsqueeze(-1))).abs().max()) == 0.0


@pytest.mark.skipif(not os.path.isdir("/root/reference/layers"), reason="needs the reference checkout")
def test_edge_gnn_matches_the_live_reference_on_random_graphs():
    """Seeded random graphs, sizes, degrees and weights: this package's Edge-GNN and the reference's (imported through
    compat with its torch >= 2 fix) on the same state_dict — forward outputs AND parameter gradients."""
    import subprocess, sys
    code = r"""
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, "/root/reference")
from categoricalnf_amd import compat
from categoricalnf_amd.layers.networks import edge_gnn as mine
from tests.test_host_cpu import _build_edge_gnn
gl = compat.reference_module("layers.networks.graph_layers")
worst = 0.0
for seed in range(6):
    g = torch.Generator().manual_seed(seed)
    V = int(torch.randint(3, 14, (1,), generator=g)); B = int(torch.randint(1, 5, (1,), generator=g))
    m = dict(step=1 + seed %% 2, hidden_nodes=16 * (1 + seed %% 3), hidden_edges=8 * (1 + seed %% 2), layers=1 + seed %% 3, max_neighbours=4,
             c_out_nodes=10, c_out_edges=6)
    torch.manual_seed(seed)
    ref = _build_edge_gnn(gl, m, 3, 2)
    for p in ref.parameters():
        p.data.normal_(0, 0.4)
    net = _build_edge_gnn(mine, m, 3, 2)
    net.load_state_dict(ref.state_dict(), strict=True)
    x1, x2 = torch.triu_indices(V, V, offset=1)
    length = torch.randint(1, V + 1, (B,), generator=g); length[0] = V
    pad = (torch.arange(V)[None] < length[:, None]).float().unsqueeze(-1)
    adj = torch.triu((torch.rand(B, V, V, generator=g) < 0.35).long(), 1)
    adj = (adj + adj.transpose(1, 2)) * (pad * pad.transpose(1, 2)).long()
    use_adj = seed %% 3 != 0
    mask_valid = adj[:, x1, x2].float() if use_adj else pad[:, x1, 0] * pad[:, x2, 0] * (torch.rand(B, x1.numel(), generator=g) > 0.4).float()
    zn = torch.randn(B, V, 3, generator=g) * pad; ze = torch.randn(B, x1.numel(), 2, generator=g) * mask_valid.unsqueeze(-1)
    wn, we = torch.randn(B, V, 10, generator=g), torch.randn(B, x1.numel(), 6, generator=g)
    outs = []
    for mod in (ref, net):
        mod.zero_grad()
        on, oe = mod(zn, ze, length=length, x_indices=(x1, x2), mask_valid=mask_valid, channel_padding_mask=pad, binary_adjacency=adj if use_adj else None)
        ((on * wn).sum() + (oe * we).sum()).backward()
        outs.append((on.detach(), oe.detach(), {k: p.grad.clone() for k, p in mod.named_parameters() if p.grad is not None}))
    (rn, re, rg), (mn, me, mg) = outs
    worst = max(worst, float((rn - mn).abs().max() / rn.abs().max().clamp(min=1)), float((re - me).abs().max() / re.abs().max().clamp(min=1)))
    assert set(rg) == set(mg)
    for k in rg:
        worst = max(worst, float((rg[k] - mg[k]).abs().max() / rg[k].abs().max().clamp(min=1)))
assert worst <= 5e-5, worst
print("EDGE GNN LIVE OK", worst)
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg"))
    assert r.returncode == 0 and "EDGE GNN LIVE OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
