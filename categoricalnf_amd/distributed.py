"""One process per GPU, batch sharded on dim 0, ONE collective on the data path.

The hot path is independent per sample, so ranks never exchange latents; the only
exchange is the all-reduce (sum) of two fp64 scalars (sum of per-sample NLL, sample count) from
which every rank gets the mean NLL / bits-per-dim — the MI355X-native replacement of the
reference's `nn.DataParallel` gather (general/train.py:36-44).  Backend "nccl" is RCCL over xGMI on
ROCm; "gloo" is used by the CPU tests."""
import os

import numpy as np
import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process => (0, 0, 1))."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend=None):
    """Initialise torch.distributed from the environment if WORLD_SIZE > 1; returns (rank, local_rank, world)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            # one process per GPU.  The device is made current BEFORE the group exists (nothing is ever allocated on
            # cuda:0 by a rank that owns another device), and the group is bound to it (`device_id`): RCCL then creates its
            # communicator eagerly on that device instead of at the first collective on whatever device is current, and a
            # barrier needs no device guess.
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        except TypeError:                            # a torch without the device_id keyword
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_bounds(total_rows, rank, world):
    """Contiguous, near-even split of `total_rows` samples: rows [lo, hi) belong to `rank`."""
    base, rem = divmod(total_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_nll(sums):
    """sums: fp64 tensor [2] = (sum of per-sample NLL, sample count) of this rank, on the device the
    backend expects.  Returns (mean NLL, bits/dim) over ALL ranks (general/task.py:139-149)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    total, count = float(sums[0].item()), float(sums[1].item())
    mean = total / max(1e-5, count)
    return mean, float(np.log2(np.exp(1)) * mean)


_SYNC_DATA_INIT = False


def sync_data_init(enable=True):
    """Opt in to (or out of) cross-rank statistics in ActNorm's data-dependent initialisation (SURVEY.md section 8e;
    layers/flows/activation_normalization.py:55-73).  The reference initialises from ONE batch on ONE process; under one
    process per GPU every rank must end up with the same bias / scales.  This package's drivers get that by feeding
    every rank the same initialisation batches; a caller whose ranks hold DIFFERENT shards of the initialisation data
    switches this on, and every ActNorm / ExtActNorm layer then all-reduces its per-channel sums — (sum x, count), then
    sum (x - mean)^2, two collectives of D + 1 fp64 per layer — so that all ranks compute the statistics of the union.
    Returns the previous setting."""
    global _SYNC_DATA_INIT
    prev, _SYNC_DATA_INIT = _SYNC_DATA_INIT, bool(enable)
    return prev


def allreduce_init_stats(acc):
    """Sum the fp64 statistics vector `acc` over the ranks, in place, when sync_data_init() is on and a process group
    with more than one rank exists; otherwise a no-op.  Called by ops.actnorm_data_init after each of its two passes."""
    if _SYNC_DATA_INIT and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    return acc


def wrap_ddp(model, device, bucket_cap_mb=25):
    """Data-parallel training wrapper replacing the reference's nn.DataParallel (general/mutils.py:243-249):
    one process per GPU, gradients all-reduced by RCCL in buckets that overlap with the HIP backward kernels.
    xGMI is point-to-point (a ring is per-link bound at ~153 GB/s), so the ~39 MB of the default set model's
    gradients go out as two ~25 MB buckets, large enough to be bandwidth- rather than latency-bound."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return model
    from torch.nn.parallel import DistributedDataParallel as DDP
    if torch.device(device).type == "cuda":
        return DDP(model, device_ids=[torch.device(device).index], bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True)
    return DDP(model, bucket_cap_mb=bucket_cap_mb)
