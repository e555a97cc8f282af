"""Two-rank data-parallel training step of the set-modelling flow (HIP forward and backward kernels under
torch DistributedDataParallel) against the same step in one process on the whole batch.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
        tools/ddp_check.py [--backend gloo|nccl] [--share-device]
--share-device puts every rank on cuda:0 (1-GPU box, gloo); on a multi-GPU node use --backend nccl (RCCL)."""
import argparse, contextlib, io, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd.distributed import init_process_group, shard_bounds, wrap_ddp
from categoricalnf_amd.experiments.set_modeling import FlowSetModeling, SetShufflingDataset
from categoricalnf_amd import functional as Fn

ap = argparse.ArgumentParser()
ap.add_argument("--backend", default="gloo")
ap.add_argument("--share-device", action="store_true")
args = ap.parse_args()
rank, local_rank, world = init_process_group(args.backend)
dev = torch.device("cuda", 0 if args.share_device else local_rank)
torch.cuda.set_device(dev)
params = lambda: {"set_size": 16, "coupling_hidden_layers": 1, "coupling_hidden_size": 32, "coupling_num_flows": 2, "coupling_mask_ratio": 0.5,
                  "coupling_num_mixtures": 8, "categ_encoding": {"use_dequantization": False, "use_variational": False, "use_decoder": False,
                                                                 "num_dimensions": 4, "flow_config": {"num_flows": 0}, "decoder_config": {}}}


def make():
    torch.manual_seed(0)
    np.random.seed(0)            # the 1x1 convolutions draw their initial rotation from numpy
    with contextlib.redirect_stdout(io.StringIO()):
        m = FlowSetModeling(params(), SetShufflingDataset)
    for p in m.parameters():
        p.data = p.data + 0.05 * torch.randn(p.shape)
    return m.to(dev).train()


B = 64
rng = np.random.RandomState(5)
x = torch.from_numpy(np.stack([rng.permutation(16) for _ in range(B)])).long().to(dev)
u = torch.rand(B * 16, 1, 4, generator=torch.Generator().manual_seed(9)).to(dev)      # encoder noise, shared by both runs
ln = torch.full((B,), 16, dtype=torch.long, device=dev)


def loss_of(model, lo, hi):
    z, ldj = model(x[lo:hi], reverse=False, length=ln[lo:hi], beta=1, noise=u[lo * 16:hi * 16])
    return Fn.PriorNllFn.apply(z, ldj, ln[lo:hi], None).sum() / B

# data-parallel: every rank its shard; DDP averages the gradients, so scale by world to get the sum
ddp = wrap_ddp(make(), dev)
lo, hi = shard_bounds(B, rank, world)
(loss_of(ddp, lo, hi) * world).backward()
torch.cuda.synchronize(dev)
ok = True
if rank == 0:
    ref = make()
    loss_of(ref, 0, B).backward()
    worst = 0.0
    inner = ddp.module if hasattr(ddp, "module") else ddp
    for (n, p), (_, q) in zip(inner.named_parameters(), ref.named_parameters()):
        if q.grad is None:
            assert p.grad is None, n
            continue
        err = (p.grad - q.grad).abs().max().item() / (q.grad.abs().max().item() + 1e-6)
        worst = max(worst, err)
    ok = worst < 2e-4
    if not ok:
        pass
    print("DDP_CHECK %s world=%d backend=%s worst relative gradient difference %.2e" % ("OK" if ok else "FAIL", world, args.backend, worst), flush=True)
if world > 1:
    t = torch.full((4,), float(rank + 1), device=dev)
    dist.all_reduce(t)
    if rank == 0:
        print("all_reduce check (expect %d):" % (world * (world + 1) // 2), t.tolist(), flush=True)
        if not ok:
            inner = ddp.module if hasattr(ddp, "module") else ddp
            for (n, p), (_, q) in list(zip(inner.named_parameters(), ref.named_parameters()))[:12]:
                if q.grad is not None:
                    print("  %-60s ddp %.4e ref %.4e" % (n, p.grad.abs().max().item(), q.grad.abs().max().item()))
    dist.barrier()
    dist.destroy_process_group()
sys.exit(0 if ok else 1)
