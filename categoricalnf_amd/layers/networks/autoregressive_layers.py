"""Autoregressive coupling sub-network of the language-modelling flow: an LSTM over the sequence whose features
feed a masked (channel-autoregressive) MLP head.

Interface, parameter names and arithmetic of layers/networks/autoregressive_layers.py (InputDropout :14-34,
TimeConcat :37-52, LSTMFeatureModel :55-97, AutoregressiveLSTMModel :100-126, AutoregFeedforward :129-184), so a
reference checkpoint loads unchanged.  Dense GEMM / LSTM work: plain PyTorch-ROCm (MIOpen / hipBLASLt); its output
is the `nn_out` the mixture-CDF kernels consume.  Differences: the connectivity masks are built with index
arithmetic instead of nested slice loops and applied functionally (`F.linear(x, W * mask)`) instead of in place on
`weight.data`, and the padded LSTM run needs no host sync (help_layers.run_padded_LSTM)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...host_utils import create_T_one_hot
from .help_layers import run_padded_LSTM


class InputDropout(nn.Module):
    """Training only: drops whole input vectors (all channels of a position) with probability `dp_rate`."""

    def __init__(self, dp_rate=0.0, scale_features=False):
        super().__init__()
        self.dp_rate = dp_rate
        self.scale_features = scale_features

    def forward(self, x):
        if not self.training:
            return x
        dropped = x.new_zeros(x.size(0), x.size(1), 1).bernoulli_(p=self.dp_rate)
        x = x * (1 - dropped)
        if self.scale_features:
            x = x * 1.0 / (1.0 - self.dp_rate)
        return x


class TimeConcat(nn.Module):
    """Appends an embedding of (position, distance to the sequence end) to every input vector."""

    def __init__(self, time_embed, input_dp_rate=0.0):
        super().__init__()
        self.time_embed_layer = time_embed
        self.input_dropout = InputDropout(input_dp_rate)

    def forward(self, x, time_embed=None, length_one_hot=None, length=None):
        if time_embed is None:
            if length_one_hot is None and length is not None:
                length_one_hot = create_T_one_hot(length, dataset_max_len=int(self.time_embed_layer.weight.shape[1] // 2), max_len=x.size(1))
            time_embed = self.time_embed_layer(length_one_hot)
        return torch.cat([self.input_dropout(x), time_embed], dim=-1)


class AutoregFeedforward(nn.Module):
    """Head with channel-autoregressive connectivity: the outputs of channel i see the LSTM features and the raw
    inputs of channels < i only.  Three linear layers whose weights are multiplied by fixed 0/1 masks."""

    def __init__(self, c_in, c_out_per_in, hidden_size, c_offset=0):
        super().__init__()
        self.c_in = c_in
        self.c_autoreg = c_in - 1 - c_offset
        self.c_out_per_in = c_out_per_in
        self.c_offset = c_offset
        self.hidden_size = hidden_size
        self.embed_size = min(max(1, int(hidden_size * 9.0 / 16.0 / (self.c_in - 1))), 96)
        self.hidden_dim_2 = int(hidden_size // 2)
        self.act_fn_1 = nn.GELU()
        self.act_fn_2 = nn.GELU()
        self.in_to_features = nn.Linear((self.c_in - 1) * 3, self.embed_size * (self.c_in - 1))
        self.features_to_hidden = nn.Linear(hidden_size + self.embed_size * (self.c_in - 1), self.hidden_dim_2 * self.c_in)
        self.hidden_to_out = nn.Linear(self.hidden_dim_2 * self.c_in, c_out_per_in * self.c_in)
        m1, m2, m3 = self._create_masks()
        self.register_buffer("mask_in_to_features", m1)
        self.register_buffer("mask_features_to_hidden", m2)
        self.register_buffer("mask_hidden_to_out", m3)

    def _create_masks(self):
        E, H2, C, O, off = self.embed_size, self.hidden_dim_2, self.c_in, self.c_out_per_in, self.c_offset
        # in_to_features [E*(C-1), 3*(C-1)]: feature block b reads the three encodings of input channel b only;
        # channels below the offset stay fully connected (:158-162)
        row_blk = torch.arange(E * (C - 1)) // E
        col_blk = torch.arange(3 * (C - 1)) // 3
        same = row_blk.view(-1, 1) == col_blk.view(1, -1)
        free = (row_blk.view(-1, 1) < off) & (col_blk.view(1, -1) < off)
        m1 = (same | free).float()
        # features_to_hidden [H2*C, hidden + E*(C-1)]: hidden block i reads all LSTM features and the feature
        # blocks of channels < max(i, offset) (:164-166)
        hid_blk = torch.arange(H2 * C) // H2
        col = torch.arange(self.hidden_size + E * (C - 1))
        limit = self.hidden_size + E * (off + (hid_blk - off).clamp(min=0))
        m2 = (col.view(1, -1) < limit.view(-1, 1)).float()
        # hidden_to_out [O*C, H2*C]: block diagonal (:168-171)
        m3 = ((torch.arange(O * C) // O).view(-1, 1) == (torch.arange(H2 * C) // H2).view(1, -1)).float()
        return m1, m2, m3

    def forward(self, features, _inps):
        if _inps.size(-1) == self.c_in:
            _inps = _inps[..., :-1]                     # the last channel conditions nothing
        enc = torch.stack([_inps, F.elu(_inps), F.elu(-_inps)], dim=-1).flatten(start_dim=-2)
        h = self.act_fn_1(F.linear(enc, self.in_to_features.weight * self.mask_in_to_features, self.in_to_features.bias))
        h = torch.cat([features, h], dim=-1)
        h = self.act_fn_2(F.linear(h, self.features_to_hidden.weight * self.mask_features_to_hidden,
                                   self.features_to_hidden.bias))
        return F.linear(h, self.hidden_to_out.weight * self.mask_hidden_to_out, self.hidden_to_out.bias)


class LSTMFeatureModel(nn.Module):

    def __init__(self, c_in, c_out, hidden_size, max_seq_len, num_layers=1, dp_rate=0.0, input_dp_rate=0.0, **kwargs):
        super().__init__()
        time_embed = nn.Linear(2 * max_seq_len, int(hidden_size // 8))
        time_embed_dim = time_embed.weight.shape[0]
        self.time_concat = TimeConcat(time_embed=time_embed, input_dp_rate=input_dp_rate)
        inp_embed_dim = hidden_size // 2 - time_embed_dim
        self.input_embed = nn.Sequential(nn.Linear(c_in, hidden_size // 2), nn.GELU(),
                                         nn.Linear(hidden_size // 2, inp_embed_dim), nn.GELU())
        self.lstm_module = nn.LSTM(input_size=inp_embed_dim + time_embed_dim, hidden_size=hidden_size,
                                   num_layers=num_layers, batch_first=True, bidirectional=False, dropout=0.0)
        self.out_layer = AutoregFeedforward(c_in=c_in, c_out_per_in=int(c_out / c_in), hidden_size=hidden_size // 2,
                                            c_offset=0)
        self.net = nn.Sequential(nn.Dropout(dp_rate), nn.Linear(hidden_size, hidden_size // 2), nn.GELU(),
                                 nn.Dropout(dp_rate))

    def forward(self, x, length=None, channel_padding_mask=None, length_one_hot=None, **kwargs):
        embed = self.time_concat(x=self.input_embed(x), length_one_hot=length_one_hot, length=length)
        # position n is conditioned on positions < n: shift the sequence right by one
        embed = torch.cat([embed.new_zeros(embed.size(0), 1, embed.size(2)), embed[:, :-1]], dim=1)
        feats = self.net(run_padded_LSTM(x=embed, lstm_cell=self.lstm_module, length=length))
        out = self.out_layer(features=feats, _inps=x)
        if channel_padding_mask is not None:
            out = out * channel_padding_mask
        return out


class AutoregressiveLSTMModel(nn.Module):

    def __init__(self, c_in, c_out, hidden_size, max_seq_len, num_layers=1, dp_rate=0.0, input_dp_rate=0.0,
                 direction=0, **kwargs):
        super().__init__()
        self.lstm_model = LSTMFeatureModel(c_in, c_out, hidden_size, num_layers=num_layers, max_seq_len=max_seq_len,
                                           dp_rate=dp_rate, input_dp_rate=input_dp_rate)
        self.reverse = (direction == 1)

    def forward(self, x, length=None, channel_padding_mask=None, **kwargs):
        if self.reverse:
            x = self._reverse_input(x, length, channel_padding_mask)
        x = self.lstm_model(x, length=length, channel_padding_mask=channel_padding_mask, shift_by_one=True, **kwargs)
        if self.reverse:
            x = self._reverse_input(x, length, channel_padding_mask)
        return x

    def _reverse_input(self, x, length, channel_padding_mask):
        """Flip every sequence inside its own length (padding stays at the end)."""
        pos = torch.arange(x.size(1), device=length.device).view(1, -1)
        src = ((length.long().view(-1, 1) - 1) - pos).clamp(min=0)
        flipped = x.gather(index=src.unsqueeze(dim=-1).expand(-1, -1, x.size(2)), dim=1)
        return flipped * channel_padding_mask
