"""Host side of the graph-colouring experiment: the dataset files, the length-bucketed batch sampler and the
validity metric of generated colourings.

Interfaces followed: experiments/graph_coloring/datasets/graph_coloring.py:18-160 (`GraphColoringDataset`: file names,
class-level arrays, `__getitem__` -> (nodes, adjacency, length), `evaluate_generations`, `num_node_types`,
`set_dataset`, `get_sampler`) and experiments/graph_coloring/datasets/mutils.py:9-61 (`BucketSampler`).  The data files
(`graph_coloring_compressed_<colors><prefix>.npz` with `nodes [G, Nmax]` (-1 = padding) and `adjacency [G, Nmax, Nmax]`,
`graph_coloring_dataidx_<colors><prefix>.npz` with `train_idx / val_idx / test_idx`) are the reference's own; they are
not shipped (no network), the loader asserts with the reference's message when they are absent;
`generate_planted_dataset` writes a synthetic set in the same two-file format (random graphs with a planted colouring).

The validity check is one vectorised pass on whatever device the samples live on (the reference loops over graphs in
numpy): a colouring is valid iff no edge joins two nodes of the same colour inside the first `length` nodes.
"""
import os
import random

import numpy as np
import torch
import torch.utils.data as data


def coloring_validity(nodes, adjacency, length=None):
    """[B] bool: graph_coloring.py:128-135 for a whole batch.

    With colours numbered from 1, `labeled[i, j] = adjacency[i, j] * colour[j]` must differ from `colour[i]` for
    every pair i, j < length (a zero adjacency entry never collides because colours start at 1)."""
    nodes = torch.as_tensor(nodes)
    adjacency = torch.as_tensor(adjacency).to(nodes.device)
    B, N = nodes.shape[0], nodes.shape[1]
    colour = nodes.long() + 1
    clash = adjacency.long() * colour[:, None, :] == colour[:, :, None]
    if length is not None:
        inside = torch.arange(N, device=nodes.device)[None, :] < torch.as_tensor(length).to(nodes.device).long()[:, None]
        clash = clash & inside[:, :, None] & inside[:, None, :]
    return ~clash.reshape(B, -1).any(dim=1)


class BucketSampler(data.Sampler):
    """Index stream in which consecutive groups of `batch_size` graphs have (nearly) the same node count; wrapped in a
    `BatchSampler` by `GraphColoringDataset.get_sampler`.  Same draws from `np.random` as mutils.py:27-57 — one
    permutation per length bucket, then per batch one bucket drawn in proportion to what is left, topped up from the
    following buckets (cyclically) when it runs dry — so that a seeded run visits the graphs in the reference's order."""

    def __init__(self, dataset, batch_size, len_step=1):
        super().__init__()
        self.dataset, self.batch_size, self.len_step = dataset, batch_size, len_step
        idx = dataset.data_indices
        node_count = (type(dataset).DATASET_NODES[idx] >= 0).sum(axis=-1) // len_step
        order = np.arange(idx.shape[0]).astype(np.int32)
        self.unique_lengths = np.unique(node_count)
        self.indices_by_lengths = [order[node_count == n] for n in self.unique_lengths]

    def __len__(self):
        return len(self.dataset)

    def __iter__(self):
        left = [np.random.permutation(bucket) for bucket in self.indices_by_lengths]
        nb, total, out = len(left), len(self), []
        while len(out) < total:
            sizes = [b.shape[0] for b in left]
            weights = [s * 1.0 / sum(sizes) for s in sizes]
            k = np.random.choice(nb, p=weights, size=1)[0]
            batch = []
            while True:
                want = self.batch_size - len(batch)
                batch += left[k][:want].tolist()
                left[k] = left[k][want:] if left[k].shape[0] > want else np.array([])
                if len(batch) >= self.batch_size or all(b.shape[0] == 0 for b in left):
                    break
                k = (k + 1) % nb
            out += batch
        return iter(out)


class GraphColoringDataset(data.Dataset):
    DATASET_NODES = None
    DATASET_ADJACENCIES = None
    DATASET_TRAIN_IDX = None
    DATASET_VAL_IDX = None
    DATASET_TEST_IDX = None
    NUM_COLORS = 3
    PREFIX = "_large"
    DATA_FILENAME = "graph_coloring_compressed_3_large.npz"
    IDX_FILENAME = "graph_coloring_dataidx_3_large.npz"

    def __init__(self, num_colors=3, train=False, val=False, test=False, order_graphs="none", data_root="data/"):
        super().__init__()
        self.train, self.val, self.test = train, val, test
        self.num_colors = num_colors
        type(self).load_dataset(data_root=data_root)
        cls = type(self)
        self.data_indices = cls.DATASET_TRAIN_IDX if train else (cls.DATASET_VAL_IDX if val else cls.DATASET_TEST_IDX)
        assert order_graphs in ["none", "rand", "largest_first", "smallest_first"], \
            "[!] ERROR: Order \"%s\" unknown" % order_graphs
        self.order_graphs = order_graphs
        print("Num %s examples: %i" % ("training" if train else ("validation" if val else "testing"),
                                       self.data_indices.shape[0]))

    def __len__(self):
        return self.data_indices.shape[0]

    def __getitem__(self, idx):
        cls = type(self)
        g = self.data_indices[idx]
        nodes = cls.DATASET_NODES[g].astype(np.int64)
        adjacency = cls.DATASET_ADJACENCIES[g].astype(np.int64)
        length = (nodes >= 0).sum().astype(np.int64)
        if self.train:                                    # colour permutation as augmentation (:56-60), `random` stream
            colours = list(range(self.num_colors))
            random.shuffle(colours)
            nodes[:length] = np.array(colours)[nodes[:length]]
        nodes = nodes + (nodes == -1)                     # padding -> colour 0
        adjacency = adjacency + (adjacency == -1)
        pos = self._node_order(adjacency, int(length), nodes.shape[0])
        if pos is not None:
            nodes, adjacency = nodes[pos], adjacency[pos, :][:, pos]
        return nodes, adjacency, length

    def _node_order(self, adjacency, length, n_max):
        """Node permutation of :62-84 (same `random` / `np.random` draws), None for the stored order."""
        if self.order_graphs == "rand":
            head = list(range(length))
            random.shuffle(head)
            return np.array(head + list(range(length, n_max)))
        if self.order_graphs in ("largest_first", "smallest_first"):
            degree = (adjacency > 0).astype(np.float32).sum(axis=1)
            if self.order_graphs == "smallest_first":
                degree = degree + (degree == 0) * 100       # padding rows last
            degree = degree + np.random.uniform(size=degree.shape) * 1e-2
            order = np.argsort(degree)
            return order[::-1] if self.order_graphs == "largest_first" else order
        return None

    @classmethod
    def load_dataset(cls, data_root="data/"):
        if cls.DATASET_NODES is None or cls.DATASET_ADJACENCIES is None:
            print("Loading graph coloring dataset (prefix=%s, %i colors)..." % (cls.PREFIX, cls.NUM_COLORS))
            path = os.path.join(data_root, cls.DATA_FILENAME)
            assert os.path.isfile(path), \
                "[!] ERROR: The graph coloring dataset could not be loaded due to a missing file.\n" + \
                "Make sure that the data is placed at: \"%s\"" % str(path)
            arr = np.load(path)
            cls.DATASET_NODES, cls.DATASET_ADJACENCIES = arr["nodes"], arr["adjacency"]
            print("Dataset loaded")
        if cls.DATASET_VAL_IDX is None:
            arr = np.load(os.path.join(data_root, cls.IDX_FILENAME))
            cls.DATASET_TRAIN_IDX, cls.DATASET_VAL_IDX, cls.DATASET_TEST_IDX = arr["train_idx"], arr["val_idx"], arr["test_idx"]

    @classmethod
    def set_dataset(cls, prefix="_tiny", num_colors=3):
        cls.PREFIX, cls.NUM_COLORS = prefix, num_colors
        cls.DATA_FILENAME = "graph_coloring_compressed_%i%s.npz" % (num_colors, prefix)
        cls.IDX_FILENAME = "graph_coloring_dataidx_%i%s.npz" % (num_colors, prefix)

    @classmethod
    def num_node_types(cls):
        return cls.NUM_COLORS

    @staticmethod
    def evaluate_generations(nodes, adjacency, length=None, **kwargs):
        """{"valid_ratio": share of valid colourings} (:114-125); tensors stay on their device."""
        valid = coloring_validity(nodes, adjacency, length)
        return {"valid_ratio": float(valid.double().mean().item())}

    def get_sampler(self, batch_size, drop_last=False, **kwargs):
        return data.BatchSampler(BucketSampler(self, batch_size, len_step=1), batch_size, drop_last=drop_last)


def generate_planted_dataset(data_root, prefix="_tiny", num_colors=3, num_graphs=60000, n_min=10, n_max=20, mean_degree=3.5,
                             val_fraction=0.1, test_fraction=0.1, seed=0):
    """Write a synthetic data set in the reference's two-file format (the published files are not reachable from here):
    random graphs with a PLANTED colouring — every node gets a random colour and edges are drawn only between nodes of
    different colours, with probability `mean_degree / n` per pair — so every graph is `num_colors`-colourable and the
    stored colouring is valid.  (The reference's own generator samples hard instances and solves them with a CSP solver,
    datasets/graph_coloring_generation.py; the statistics differ, the file format and the task do not.)  Isolated nodes
    get one edge to a node of another colour so that no node is unconstrained."""
    rng = np.random.RandomState(seed)
    nodes = -np.ones((num_graphs, n_max), dtype=np.int8)
    adjacency = -np.ones((num_graphs, n_max, n_max), dtype=np.int8)
    for g in range(num_graphs):
        n = rng.randint(n_min, n_max + 1)
        colours = rng.randint(0, num_colors, size=n)
        if len(set(colours.tolist())) < 2:
            colours[0], colours[1] = 0, 1
        differ = colours[:, None] != colours[None, :]
        a = np.triu((rng.rand(n, n) < mean_degree / n) & differ, 1)
        a = a | a.T
        for i in np.where(a.sum(1) == 0)[0]:
            j = rng.choice(np.where(differ[i])[0])
            a[i, j] = a[j, i] = True
        nodes[g, :n] = colours
        adjacency[g, :n, :n] = a
    os.makedirs(data_root, exist_ok=True)
    order = rng.permutation(num_graphs)
    n_val, n_test = int(num_graphs * val_fraction), int(num_graphs * test_fraction)
    np.savez_compressed(os.path.join(data_root, "graph_coloring_compressed_%i%s.npz" % (num_colors, prefix)),
                        nodes=nodes, adjacency=adjacency)
    np.savez_compressed(os.path.join(data_root, "graph_coloring_dataidx_%i%s.npz" % (num_colors, prefix)),
                        train_idx=order[n_val + n_test:], val_idx=order[:n_val], test_idx=order[n_val:n_val + n_test])
    return nodes, adjacency
