"""Coupling-subnet helpers that stay plain PyTorch-ROCm (dense GEMMs -> hipBLASLt / MFMA).

Mirrors the names of layers/networks/help_layers.py in the reference: SimpleLinearLayer (:57-73),
LinearNet (:76-107) and run_sequential_with_mask (:111-124).  Parameter names are kept so that the
reference's checkpoints load (`…pred_net.layer.weight`, `…nn.inp_layer.0.weight`, …)."""
import numpy as np
import torch
import torch.nn as nn


class SimpleLinearLayer(nn.Module):
    """Embedding features -> (bias, scale) of an ExtActNorm; one Linear."""

    def __init__(self, c_in, c_out, data_init=False):
        super().__init__()
        self.layer = nn.Linear(c_in, c_out)
        if data_init:
            half = int(c_out // 2)
            with torch.no_grad():
                self.layer.weight[half:, :] = 0                      # scales start at tanh(0) = 0
                self.layer.weight.mul_(4 / np.sqrt(c_out / 2))       # spread of the class means
                self.layer.bias.zero_()

    def forward(self, x, **kwargs):
        return self.layer(x)

    def initialize_zeros(self):
        with torch.no_grad():
            self.layer.weight.zero_()
            self.layer.bias.zero_()


class LinearNet(nn.Module):
    """MLP used as coupling subnet inside the linear-flow encoder; optional conditioning features."""

    def __init__(self, c_in, c_out, num_layers, hidden_size, ext_input_dims=0, zero_init=False):
        super().__init__()
        self.inp_layer = nn.Sequential(nn.Linear(c_in, hidden_size), nn.GELU())
        blocks = []
        for i in range(num_layers):
            blocks += [nn.Linear(hidden_size + (ext_input_dims if i == 0 else 0), hidden_size), nn.GELU()]
        blocks.append(nn.Linear(hidden_size, c_out))
        self.main_net = nn.Sequential(*blocks)
        if zero_init:
            with torch.no_grad():
                self.main_net[-1].weight.zero_()
                self.main_net[-1].bias.zero_()

    def forward(self, x, ext_input=None, **kwargs):
        feat = self.inp_layer(x)
        if ext_input is not None:
            feat = torch.cat([feat, ext_input], dim=-1)
        return self.main_net(feat)

    def set_bias(self, bias):
        # the reference assigns the tensor as it comes (help_layers.py:106); a float64 numpy prior then makes the bias
        # double, which torch >= 2 refuses to mix with float activations — keep the parameter's own dtype
        last = self.main_net[-1].bias
        last.data = bias.to(device=last.device, dtype=last.dtype)


def run_sequential_with_mask(net, x, length=None, channel_padding_mask=None, src_key_padding_mask=None,
                             length_one_hot=None, time_embed=None, gt=None, importance_weight=1,
                             detail_out=False, **kwargs):
    """Run an nn.Sequential subnet, zeroing padded positions before and after (help_layers.py:111-124)."""
    if channel_padding_mask is None:
        out = net(x)
    else:
        h = x * channel_padding_mask
        for layer in net:
            h = layer(h)
        out = h * channel_padding_mask
    return (out, dict()) if detail_out else out


def run_padded_LSTM(x, lstm_cell, length, input_memory=None, return_final_states=False):
    """layers/networks/help_layers.py:127-142.  The reference sorts by length, packs, runs the LSTM and pads back
    (pack_padded_sequence wants the lengths on the host: one device sync per call).  Every caller here uses a
    unidirectional LSTM, whose output at a position depends on earlier positions only, so running it on the padded
    batch and zeroing the positions past each length gives the same tensor without the sync."""
    if getattr(lstm_cell, "bidirectional", False):
        raise NotImplementedError("run_padded_LSTM: unidirectional LSTMs only")
    outputs, _ = lstm_cell(x, input_memory)
    if length is not None:
        keep = torch.arange(x.size(1), device=x.device).view(1, -1) < length.view(-1, 1)
        outputs = outputs * keep.unsqueeze(dim=-1).to(outputs.dtype)
    return outputs
