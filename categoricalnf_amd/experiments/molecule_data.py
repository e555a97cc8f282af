"""A synthetic molecule-LIKE graph set in the file format of the reference's Zinc250k data set
(experiments/molecule_generation/datasets/zinc250k.py:26-35, 66-79: `zinc250k/zinc250k_compressed.npz` with `nodes`
int8 [M, 38], -1 = padding, atom-type index 0..8, and `adjacency` int8 [M, 38, 38], 0 = no bond, 1..3 = bond order;
`zinc250k/zinc250k_dataidx.npz` with `train_idx` / `val_idx`, of which the first 8192 validation graphs are the validation
set and the rest the test set).

The published files are not reachable from here and nothing in this image can judge chemistry (no RDKit), so these graphs
make no claim to be molecules: random trees of 8..38 atoms with a few ring closures, at most four bonds per atom, a
carbon-heavy type distribution and mostly single bonds — the sizes, sparsity and category counts the molecule flow
(configs[4]: node types 9 / D = 6 / K = 16, edge types 3 + virtual edges / D = 2 / K = 8) is built for.  They let the
reference's own `experiments/molecule_generation/train.py` (through `categoricalnf_amd.run_reference`) and this package's
`GraphCNF` be driven end to end where the real data is missing."""
import os

import numpy as np

MAX_NODES = 38
NUM_NODE_TYPES = 9
TYPE_PROBS = np.array([0.74, 0.12, 0.10, 0.014, 0.001, 0.018, 0.007, 0.002, 0.0005])


def random_molecule_like_graph(rng, n):
    """(types int [n], bond orders int [n, n]) of one graph: a random tree with up to two ring closures."""
    types = rng.choice(NUM_NODE_TYPES, size=n, p=TYPE_PROBS / TYPE_PROBS.sum())
    bonds = np.zeros((n, n), dtype=np.int8)
    degree = np.zeros(n, dtype=np.int64)
    for i in range(1, n):
        j = rng.choice(np.where(degree[:i] < 3)[0])              # a node that can still take a neighbour
        bonds[i, j] = bonds[j, i] = 1 if rng.rand() < 0.85 else 2
        degree[i] += 1
        degree[j] += 1
    for _ in range(rng.randint(0, 3)):
        i, j = rng.randint(0, n, size=2)
        if i != j and bonds[i, j] == 0 and degree[i] < 4 and degree[j] < 4:
            bonds[i, j] = bonds[j, i] = 1
            degree[i] += 1
            degree[j] += 1
    return types, bonds


def generate_molecule_like_dataset(data_root, num_graphs=12000, n_min=8, n_max=MAX_NODES, num_val=8192 + 256, seed=0):
    """Write the two files under `data_root`/zinc250k/ and return (nodes, adjacency).  `num_val` must exceed 8192 for the
    reference's test split (`val_idx[8192:]`) to be non-empty."""
    if num_val >= num_graphs:
        raise ValueError("num_val must be smaller than num_graphs")
    rng = np.random.RandomState(seed)
    nodes = -np.ones((num_graphs, MAX_NODES), dtype=np.int8)
    adjacency = np.zeros((num_graphs, MAX_NODES, MAX_NODES), dtype=np.int8)
    for g in range(num_graphs):
        n = rng.randint(n_min, n_max + 1)
        nodes[g, :n], adjacency[g, :n, :n] = random_molecule_like_graph(rng, n)
    os.makedirs(os.path.join(data_root, "zinc250k"), exist_ok=True)
    order = rng.permutation(num_graphs)
    np.savez_compressed(os.path.join(data_root, "zinc250k", "zinc250k_compressed.npz"), nodes=nodes, adjacency=adjacency)
    np.savez_compressed(os.path.join(data_root, "zinc250k", "zinc250k_dataidx.npz"), train_idx=order[num_val:],
                        val_idx=order[:num_val])
    return nodes, adjacency
