"""Data-dependent ActNorm initialisation with every rank holding a DIFFERENT shard of the initialisation batch
(distributed.sync_data_init): bias / scales on every rank must equal those one process computes from the whole batch
(layers/flows/activation_normalization.py:55-73).
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 \
        tools/init_sync_check.py [--backend gloo|nccl] [--share-device]"""
import argparse, contextlib, io, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd.distributed import init_process_group, shard_bounds, sync_data_init
from categoricalnf_amd.layers.flows.activation_normalization import ActNormFlow

ap = argparse.ArgumentParser()
ap.add_argument("--backend", default="gloo")
ap.add_argument("--share-device", action="store_true")
args = ap.parse_args()
rank, local_rank, world = init_process_group(args.backend)
dev = torch.device("cuda", 0 if args.share_device else local_rank)
torch.cuda.set_device(dev)
B, N, D = 96, 17, 5
g = torch.Generator().manual_seed(3)
x = (torch.randn(B, N, D, generator=g) * torch.tensor([0.5, 1.0, 2.0, 3.0, 0.1]) + torch.tensor([1.0, -2.0, 0.0, 5.0, 0.3])).to(dev)
ln = torch.randint(N // 2, N + 1, (B,), generator=g)
pad = (torch.arange(N)[None, :] < ln[:, None]).float().unsqueeze(-1).to(dev)


def init(xs, ps):
    layer = ActNormFlow(D).to(dev)
    with contextlib.redirect_stdout(io.StringIO()):
        layer.data_init_forward(xs, channel_padding_mask=ps)
    return layer.bias.data.clone(), layer.scales.data.clone()


b_all, s_all = init(x, pad)                                   # one process, the whole batch (no collective: switch is off)
lo, hi = shard_bounds(B, rank, world)                         # uneven shards are fine: counts are reduced too
b_loc, s_loc = init(x[lo:hi], pad[lo:hi])                     # per-rank statistics: what a user's own DDP loop got silently
sync_data_init(True)
b_syn, s_syn = init(x[lo:hi], pad[lo:hi])
sync_data_init(False)
err = max((b_syn - b_all).abs().max().item(), (s_syn - s_all).abs().max().item())
gap = max((b_loc - b_all).abs().max().item(), (s_loc - s_all).abs().max().item())
assert err < 1e-6, "rank %d: synchronised init differs from the whole-batch init by %g" % (rank, err)
assert world == 1 or gap > 1e-4, "the shards should not share their statistics by accident (gap %g)" % gap
print("INIT_SYNC OK rank %d of %d: |sync - whole batch| = %.1e, |per-rank - whole batch| = %.1e" % (rank, world, err, gap), flush=True)
