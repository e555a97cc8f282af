"""Model assembly for set modelling (the first caller of the hot path): categorical encoder +
n x (ActNorm, invertible 1x1 conv, mixture-CDF coupling with a Transformer sub-network).

Same structure, constructor arguments and parameter names as the reference's
experiments/set_modeling/flow_model.py (FlowSetModeling :16-102, CouplingTransformerNet :106-136), so
a reference checkpoint loads unchanged.  The Transformer sub-network is dense GEMM / attention work
and stays plain PyTorch-ROCm; every flow layer around it runs on the HIP kernels."""
import torch
import torch.nn as nn

from ..host_utils import create_channel_mask, create_transformer_mask
from ..layers.categorical_encoding.mutils import create_encoding
from ..layers.flows.activation_normalization import ActNormFlow
from ..layers.flows.coupling_layer import CouplingLayer
from ..layers.flows.flow_model import FlowModel
from ..layers.flows.mixture_cdf_layer import MixtureCDFCoupling
from ..layers.flows.permutation_layers import InvertibleConv


class CouplingTransformerNet(nn.Module):
    """Permutation-equivariant coupling sub-network: MLP in, `num_layers` Transformer encoder layers
    (4 heads, feed-forward 2x hidden, GELU, no dropout), LayerNorm + MLP out."""

    def __init__(self, c_in, c_out, num_layers, hidden_size):
        super().__init__()
        self.input_layer = nn.Sequential(nn.Linear(c_in, hidden_size), nn.GELU(), nn.Linear(hidden_size, hidden_size))
        self.transformer_layers = nn.ModuleList([
            nn.TransformerEncoderLayer(hidden_size, nhead=4, dim_feedforward=2 * hidden_size, dropout=0.0, activation='gelu')
            for _ in range(num_layers)])
        self.output_layer = nn.Sequential(nn.LayerNorm(hidden_size), nn.Linear(hidden_size, hidden_size), nn.GELU(),
                                          nn.Linear(hidden_size, c_out))

    def forward(self, x, src_key_padding_mask, **kwargs):
        h = self.input_layer(x.transpose(0, 1))          # [N, B, hidden] as nn.TransformerEncoderLayer expects
        for layer in self.transformer_layers:
            h = layer(h, src_key_padding_mask=src_key_padding_mask)
        return self.output_layer(h).transpose(0, 1)


class FlowSetModeling(FlowModel):

    def __init__(self, model_params, dataset_class):
        super().__init__(layers=None, name="Set Modeling Flow")
        self.model_params = model_params
        self.dataset_class = dataset_class
        self.set_size = self.model_params["set_size"]
        self.vocab_size = self.dataset_class.get_vocab_size(self.set_size)
        self._create_layers()
        self.print_overview()

    def _create_layers(self):
        p = self.model_params
        self.latent_dim = p["categ_encoding"]["num_dimensions"]
        model_func = lambda c_out: CouplingTransformerNet(c_in=self.latent_dim, c_out=c_out,
                                                          num_layers=p["coupling_hidden_layers"],
                                                          hidden_size=p["coupling_hidden_size"])
        p["categ_encoding"]["flow_config"]["model_func"] = model_func
        p["categ_encoding"]["flow_config"]["block_type"] = "Transformer"
        self.encoding_layer = create_encoding(p["categ_encoding"], dataset_class=self.dataset_class,
                                              vocab_size=self.vocab_size)
        if self.latent_dim > 1:
            channel = CouplingLayer.create_channel_mask(self.latent_dim, ratio=p["coupling_mask_ratio"])
            mask_of = lambda i: channel
        else:
            chess = CouplingLayer.create_chess_mask()
            mask_of = lambda i: chess if i % 2 == 0 else 1 - chess
        layers = []
        for i in range(p["coupling_num_flows"]):
            layers += [ActNormFlow(self.latent_dim), InvertibleConv(self.latent_dim),
                       MixtureCDFCoupling(c_in=self.latent_dim, mask=mask_of(i), model_func=model_func,
                                          block_type="Transformer", num_mixtures=p["coupling_num_mixtures"])]
        self.flow_layers = nn.ModuleList([self.encoding_layer] + layers)

    def forward(self, z, ldj=None, reverse=False, length=None, **kwargs):
        if length is not None:
            # max_len = the padded set size: same masks as the reference's `length.max()` for any batch it can
            # process, without the host sync (keeps the pass capturable in a HIP graph)
            kwargs["src_key_padding_mask"] = create_transformer_mask(length, max_len=z.size(1))
            kwargs["channel_padding_mask"] = create_channel_mask(length, max_len=z.size(1))
        return super().forward(z, ldj=ldj, reverse=reverse, length=length, **kwargs)

    def initialize_data_dependent(self, batch_list):
        print("Initializing data dependent...")
        with torch.no_grad():
            for _, kwargs in batch_list:
                kwargs["src_key_padding_mask"] = create_transformer_mask(kwargs["length"])
                kwargs["channel_padding_mask"] = create_channel_mask(kwargs["length"])
            for layer in self.flow_layers:
                batch_list = FlowModel.run_data_init_layer(batch_list, layer)


class SetShufflingDataset:
    """Deterministic validation / test sets of the set-shuffling task
    (experiments/set_modeling/datasets/set_shuffling.py:17-47): 32768 permutations of `set_size`
    drawn with numpy's legacy generator seeded 123 (val) / 101 (test)."""

    def __init__(self, set_size, train=True, val=False, test=False, **kwargs):
        import numpy as np
        self.set_size = set_size
        self.num_classes = set_size
        self.shuffle_set = None
        if val or test:
            rng = np.random.RandomState(123 if val else 101)
            self.shuffle_set = np.stack([rng.permutation(set_size) for _ in range(32768)])

    @staticmethod
    def get_vocab_size(set_size):
        return set_size

    @staticmethod
    def optimum_bpd(set_size):
        import numpy as np
        return float(sum(np.log2(i) for i in range(1, set_size + 1)) / set_size)

    def sample(self, batch_size, rng):
        """One training batch: `batch_size` uniformly random permutations (int64 [B, set_size])."""
        import numpy as np
        return np.stack([rng.permutation(self.set_size) for _ in range(batch_size)])

    def eval_sets(self):
        return self.shuffle_set


def bounded_partitions(total, parts, largest):
    """All ways to write `total` as `parts` integers in [1, largest], each listed once as a non-increasing
    sequence, in descending lexicographic order (the order of the reference's create_all_examples,
    experiments/set_modeling/datasets/set_summation.py:84-116, so equal seeds draw equal validation sets)."""
    out = []

    def extend(prefix, remaining, slots, cap):
        if slots == 0:
            if remaining == 0:
                out.append(prefix)
            return
        # the next part is at most `cap`, leaves >= 1 for every later slot and cannot undershoot slots * part
        for k in range(min(cap, remaining - (slots - 1)), 0, -1):
            if k * slots < remaining:
                break
            extend(prefix + [k], remaining - k, slots - 1, k)

    extend([], total, parts, largest)
    return out


class SetSummationDataset:
    """Set summation (set_summation.py:20-58): multisets of `set_size` numbers from 1..set_size that sum to `max_sum`,
    drawn with probability proportional to their number of distinct orderings and presented in random order, shifted
    to categories 0..set_size-1.  Validation / test: 32768 sets from numpy's legacy generator seeded 123 / 101 (the
    same multisets as the reference; the reference orders each of them with Python's unseeded `random.shuffle`, here
    a generator seeded 123 / 101 does, so the sets are reproducible)."""

    def __init__(self, set_size, max_sum=42, train=True, val=False, test=False, **kwargs):
        import math
        from collections import Counter
        import numpy as np
        self.set_size = set_size
        self.num_classes = set_size
        self.max_sum = max_sum
        self.multisets = np.array(bounded_partitions(max_sum, set_size, set_size), dtype=np.int64)
        orderings = np.array([math.factorial(set_size) / np.prod([math.factorial(c) for c in Counter(m.tolist()).values()])
                              for m in self.multisets])
        self.num_orderings = float(orderings.sum())
        self.probs = orderings / orderings.sum()
        self.fixed = None
        if val or test:
            np.random.seed(123 if val else 101)             # the reference seeds the global generator (:33)
            idx = np.random.choice(len(self.multisets), p=self.probs, size=(32768,))
            order = np.random.RandomState(123 if val else 101)
            self.fixed = np.stack([self.multisets[i][order.permutation(set_size)] for i in idx]) - 1

    @staticmethod
    def get_vocab_size(set_size):
        return set_size

    def optimum_bpd(self):
        import numpy as np
        return float(np.log2(self.num_orderings) / self.set_size)

    def sample(self, batch_size, rng):
        import numpy as np
        idx = rng.choice(len(self.multisets), p=self.probs, size=(batch_size,))
        return np.stack([self.multisets[i][rng.permutation(self.set_size)] for i in idx]) - 1

    def eval_sets(self):
        return self.fixed
