"""Golden vectors for categoricalnf_amd/experiments/graph_coloring_data.py, made by running the REFERENCE's
GraphColoringDataset / BucketSampler (experiments/graph_coloring/datasets/{graph_coloring,mutils}.py) on a small
synthetic graph set placed in its class attributes (the real data files are not reachable).

    PYTHONPATH=/root/reference MPLBACKEND=Agg python oracle/gen_graph_coloring_data_golden.py

Test infrastructure only: runs in the build container (needs /root/reference); the .npz it writes is what travels."""
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from experiments.graph_coloring.datasets.graph_coloring import GraphColoringDataset as Ref  # noqa: E402
from experiments.graph_coloring.datasets.mutils import BucketSampler as RefSampler          # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "graph_coloring_data.npz")


def make_graphs(G=240, n_max=12, colors=3, seed=7):
    rng = np.random.RandomState(seed)
    nodes = -np.ones((G, n_max), dtype=np.int8)
    adj = -np.ones((G, n_max, n_max), dtype=np.int8)
    for g in range(G):
        n = rng.randint(4, n_max + 1)
        col = rng.randint(0, colors, size=n)
        a = (rng.rand(n, n) < 0.35).astype(np.int8)
        a = np.triu(a, 1)
        a = a + a.T
        if g % 3 != 0:                       # two thirds: proper colourings (edges only between different colours)
            a = a * (col[:, None] != col[None, :])
        nodes[g, :n] = col
        adj[g, :n, :n] = a
    return nodes, adj


def ref_sampler(dataset, batch_size, len_step=1):
    """The reference's sampler without its constructor: `data.Sampler.__init__(dataset)` is a TypeError on torch >= 2.2
    (mutils.py:12); its attributes are set by hand and its own `_prepare` / `__iter__` do the work."""
    s = RefSampler.__new__(RefSampler)
    s.dataset, s.batch_size, s.len_step = dataset, batch_size, len_step
    s._prepare()
    return s


def main():
    nodes, adj = make_graphs()
    G = nodes.shape[0]
    perm = np.random.RandomState(1).permutation(G)
    Ref.DATASET_NODES, Ref.DATASET_ADJACENCIES = nodes, adj
    Ref.DATASET_TRAIN_IDX, Ref.DATASET_VAL_IDX, Ref.DATASET_TEST_IDX = perm[:160], perm[160:200], perm[200:]
    out = {"nodes": nodes, "adjacency": adj, "train_idx": perm[:160], "val_idx": perm[160:200], "test_idx": perm[200:]}

    # validity of every stored graph, and of a padded batch through evaluate_generations
    val = Ref(val=True)
    batch = [val[i] for i in range(len(val))]
    bn = np.stack([b[0] for b in batch]); ba = np.stack([b[1] for b in batch]); bl = np.stack([b[2] for b in batch])
    out["val_nodes"], out["val_adjacency"], out["val_length"] = bn, ba, bl
    out["val_valid"] = np.array([Ref._check_validity(bn[i], ba[i], bl[i]) for i in range(bn.shape[0])])
    out["val_valid_ratio"] = np.float64(Ref.evaluate_generations(torch.from_numpy(bn), torch.from_numpy(ba), torch.from_numpy(bl))["valid_ratio"])
    # (length=None makes the reference slice with a float under numpy >= 1.12 and stop: not pinned)

    # sampler index streams
    train = Ref(train=True)
    for bs in (16, 7):
        for seed in (0, 5):
            np.random.seed(seed)
            out["sampler_bs%d_seed%d" % (bs, seed)] = np.array(list(iter(ref_sampler(train, bs))), dtype=np.int64)
    np.random.seed(3)
    import torch.utils.data as data
    out["batch_sampler_bs16_seed3"] = np.array([b for b in data.BatchSampler(ref_sampler(train, 16), 16, drop_last=True)],
                                               dtype=np.int64)

    # __getitem__ under every node order (train split: colour shuffle from `random`)
    for order in ("none", "rand", "largest_first", "smallest_first"):
        ds = Ref(train=True, order_graphs=order)
        random.seed(11); np.random.seed(11)
        items = [ds[i] for i in range(24)]
        out["item_%s_nodes" % order] = np.stack([it[0] for it in items])
        out["item_%s_adjacency" % order] = np.stack([it[1] for it in items])
        out["item_%s_length" % order] = np.stack([it[2] for it in items])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
