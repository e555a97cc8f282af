"""Embedding layer factory and the small decoder MLP (layers/categorical_encoding/decoder.py)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...host_utils import get_param_val
from ..networks.help_layers import LinearNet


def create_embed_layer(vocab, vocab_size, default_embed_layer_dims):
    """nn.Embedding from scratch or from a torchtext vocabulary's vectors (decoder.py:12-21)."""
    use_vectors = (vocab is not None and vocab.vectors is not None)
    dims = vocab.vectors.shape[1] if use_vectors else default_embed_layer_dims
    vocab_size = len(vocab) if use_vectors else vocab_size
    embed = nn.Embedding(vocab_size, dims)
    if use_vectors:
        embed.weight.data.copy_(vocab.vectors)
        embed.weight.requires_grad = True
    return embed, vocab_size


def create_decoder(num_categories, num_dims, config, **kwargs):
    return DecoderLinear(num_categories, embed_dim=num_dims,
                         hidden_size=get_param_val(config, "hidden_size", 64),
                         num_layers=get_param_val(config, "num_layers", 1), **kwargs)


class DecoderLinear(nn.Module):
    """MLP latent -> class log-probabilities on the features [z, elu(z), elu(-z)] (decoder.py:35-63).
    Dense GEMMs: stays PyTorch-ROCm."""

    def __init__(self, num_categories, embed_dim, hidden_size, num_layers, class_prior_log=None):
        super().__init__()
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.layers = LinearNet(c_in=3 * embed_dim, c_out=num_categories, num_layers=num_layers, hidden_size=hidden_size)
        self.log_softmax = nn.LogSoftmax(dim=-1)
        if class_prior_log is not None:
            if not isinstance(class_prior_log, torch.Tensor):
                class_prior_log = torch.from_numpy(class_prior_log)
            self.layers.set_bias(class_prior_log)

    def forward(self, z_cont):
        feats = torch.cat([z_cont, F.elu(z_cont), F.elu(-z_cont)], dim=-1)
        return self.log_softmax(self.layers(feats))

    def info(self):
        return "Linear model with hidden size %i and %i layers" % (self.hidden_size, self.num_layers)
