"""Mixture-model / linear-flow categorical encoder.

Interface of layers/categorical_encoding/linear_encoding.py (LinearCategoricalEncoding :17-209,
_create_flows :214-250).  Mixture-model encoding (num_flows == 0 — the default of every
experiment) runs as ONE kernel per direction (cnf_encoder_forward / cnf_encoder_decode): the
class-conditional flow is a single ExtActNorm whose predictor sees only the class embedding, so it
is a [C, 2D] table; the kernel takes the uniform draw, turns it into logistic noise itself
(LogisticDistribution.sample fused in: cnf_encoder_forward_sampled), computes the forward push, the
per-category log-prob over all C classes, the posterior and the token log-det.
Linear-flow encoding (num_flows > 0) composes the ExtActNorm / InvertibleConv / CouplingLayer
kernels over the expanded [T*C, 1, D] tensor like the reference."""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import functional as Fn
from ... import ops
from ...host_utils import get_param_val, one_hot
from ..flows.activation_normalization import ExtActNormFlow
from ..flows.coupling_layer import CouplingLayer
from ..flows.distributions import LogisticDistribution
from ..flows.flow_layer import FlowLayer
from ..flows.permutation_layers import InvertibleConv
from ..networks.help_layers import LinearNet, SimpleLinearLayer
from .decoder import create_decoder, create_embed_layer


class LinearCategoricalEncoding(FlowLayer):

    def __init__(self, num_dimensions, flow_config, dataset_class=None, vocab=None, vocab_size=-1,
                 use_decoder=False, decoder_config=None, default_embed_layer_dims=64,
                 category_prior=None, **kwargs):
        super().__init__()
        self.use_decoder = use_decoder
        self.dataset_class = dataset_class
        self.D = num_dimensions
        self.embed_layer, self.vocab_size = create_embed_layer(vocab, vocab_size, default_embed_layer_dims)
        self.num_categories = self.vocab_size
        self.prior_distribution = LogisticDistribution(mu=0.0, sigma=1.0)
        self.flow_layers = _create_flows(num_dims=num_dimensions, embed_dims=self.embed_layer.weight.shape[1],
                                         config=flow_config)
        if self.use_decoder:
            self.decoder = create_decoder(num_categories=self.vocab_size, num_dims=self.D, config=decoder_config)
        if category_prior is None:
            category_prior = torch.zeros(self.vocab_size, dtype=torch.float32)      # uniform
        else:
            assert category_prior.shape[0] == self.num_categories, \
                "[!] ERROR: Category prior needs to be of size [%i] but is %s" % (self.num_categories, str(category_prior.shape))
            if isinstance(category_prior, np.ndarray):
                category_prior = torch.from_numpy(category_prior)
        self.register_buffer("category_prior", F.log_softmax(category_prior, dim=-1))
        # where the encoder noise is drawn: "device" (Philox on the GPU, no H2D copy) or "cpu"
        # (the reference's CPU generator, linear_encoding.py:76)
        self.noise_generator = os.environ.get("CNF_NOISE", "device")

    # ---- helpers -------------------------------------------------------------------------------
    def _is_mixture_model(self):
        """One ExtActNorm conditioned on the class only: the class-conditional flow is a [C, 2D] table."""
        return (len(self.flow_layers) == 1 and isinstance(self.flow_layers[0], ExtActNormFlow)
                and not self.flow_layers[0].make_unique and not self.use_decoder and self.D <= 16)

    def _kernel_path(self, needs_grad):
        """The one-kernel encoder for every mixture-model vocabulary: class table resident in LDS when it fits, the
        class-tiled kernels (forward, decode and backward) beyond that."""
        return self._is_mixture_model()

    def class_table(self):
        """[C, 2D] rows [bias | scales_raw] = pred_net(embed_layer(c)) for every class (one tiny GEMM)."""
        return self.flow_layers[0].pred_net(self.embed_layer.weight)

    def _noise(self, tokens, device, noise=None):
        shape = (tokens, 1, self.D)
        if noise is not None:                        # injected U[0,1) draw (parity tests)
            return self.prior_distribution.sample(shape=shape, device=device, uniform=noise.reshape(shape))
        return self.prior_distribution.sample(shape=shape, device=device, generator=self.noise_generator)

    def _uniform_draw(self, tokens, device, noise=None):
        """The U[0,1) draw behind `_noise`, as [tokens, D] on the device: the one-kernel encoder samples the logistic noise
        from it itself (the steps of LogisticDistribution.sample :105-122 up to cnf_logistic_from_uniform)."""
        shape = (tokens, 1, self.D)
        if noise is not None:                        # injected U[0,1) draw (parity tests)
            u = noise.reshape(shape)
        else:
            u = self.prior_distribution.uniform(shape, generator=self.noise_generator, device=device)
        if u.dim() == len(shape) + 1:
            u = u.squeeze(dim=-1)
        return u.to(device=device, dtype=torch.float32).reshape(tokens, self.D)

    # ---- forward -------------------------------------------------------------------------------
    def forward(self, z, ldj=None, reverse=False, beta=1, delta=0.0, channel_padding_mask=None, noise=None, **kwargs):
        batch_size, seq_length = z.size(0), z.size(1)
        detailed_ldj = {}
        if not reverse:
            table = self.class_table() if self._is_mixture_model() else None
            if table is not None and self._kernel_path(Fn.needs_grad(table)):
                u = self._uniform_draw(batch_size * seq_length, z.device, noise)
                squeeze = float(self.prior_distribution.eps)
                # FlowModel passes get_ldj_per_layer on: the monitoring scalars (five global reductions over the posterior of every
                # token) are only computed for a pass whose per-layer report is asked for — a bare call of the layer reports them as
                # the reference does
                want_stats = self.training and kwargs.get("get_ldj_per_layer", True)
                if isinstance(beta, torch.Tensor):
                    # beta in a device scalar (a captured training step: the schedule is written into it between replays)
                    z_out, ldj_loc, cpl = Fn.EncoderForwardDevBetaFn.apply(table, z, u, self.category_prior, channel_padding_mask,
                                                                           beta, squeeze)
                elif Fn.needs_grad(table):
                    z_out, ldj_loc, cpl = Fn.EncoderForwardFn.apply(table, z, u, self.category_prior, channel_padding_mask,
                                                                    float(beta), want_stats, None, squeeze)
                else:
                    z_out, ldj_loc, cpl = ops.encoder_forward(z, u, table, self.category_prior, beta=float(beta),
                                                              channel_padding_mask=channel_padding_mask,
                                                              want_class_prob=want_stats, uniform_squeeze=squeeze)
                if want_stats:
                    detailed_ldj = self._train_stats(z_out, cpl, channel_padding_mask)
            else:
                z_out, ldj_loc, detailed_ldj = self._forward_composed(z, beta, channel_padding_mask, noise)
        else:
            assert z.size(-1) == self.D, \
                "[!] ERROR in categorical decoding: Input must have %i latent dimensions but got %i" % (self.D, z.shape[-1])
            if self._kernel_path(False):
                z_out = ops.encoder_decode(z, self.class_table().detach(), self.category_prior)
            elif self.use_decoder:
                z_out = self.decoder(z.reshape(batch_size * seq_length, 1, self.D)).argmax(dim=-1).reshape(batch_size, seq_length)
            else:
                z_out = self._posterior_sample(z.reshape(batch_size * seq_length, 1, self.D)).reshape(batch_size, seq_length)
            ldj_loc = z.new_zeros(batch_size, dtype=torch.float32)
        ldj = ldj + ldj_loc if ldj is not None else ldj_loc
        return z_out, ldj, detailed_ldj

    def fusable_with_actconv(self, differentiable=False):
        """FlowModel may run this layer together with the ActNorm + 1x1 convolution behind it: the one-kernel mixture-model
        encoder in evaluation mode (training mode reports statistics of the encoder's own latents, :95-106) — or, with
        `differentiable`, on a training pass whose per-layer report nobody asked for (the statistics would be dropped)."""
        return self._is_mixture_model() and (differentiable or not self.training)

    def forward_with_actconv(self, z, act_bias, act_scales, conv_weight, conv_sldj, ldj=None, beta=1, channel_padding_mask=None,
                             length=None, noise=None, differentiable=False):
        """forward() followed by ActNormFlow.forward and InvertibleConv.forward of the next flow step, as one kernel
        (cnf_encoder_forward_actconv); returns (latents after the convolution, running log-det).  `differentiable`: through
        functional.EncoderActConvFn (one backward for the three layers)."""
        u = self._uniform_draw(z.size(0) * z.size(1), z.device, noise)
        if differentiable:
            return Fn.EncoderActConvFn.apply(self.class_table(), act_bias, act_scales, conv_weight, conv_sldj, ldj, z, u,
                                             self.category_prior, channel_padding_mask, length, float(beta),
                                             float(self.prior_distribution.eps))
        return ops.encoder_forward_actconv(z, u, self.class_table(), self.category_prior, act_bias, act_scales, conv_weight,
                                           conv_sldj, beta=float(beta), channel_padding_mask=channel_padding_mask,
                                           length=length, ldj=ldj, uniform_squeeze=float(self.prior_distribution.eps))

    def decode_with_actconv(self, z, act_bias, act_scales, conv_weight_inv, conv_sldj, ldj=None, channel_padding_mask=None,
                            length=None):
        """InvertibleConv.forward(reverse=True) and ActNormFlow.forward(reverse=True) of the first flow step followed by
        this layer's reverse pass (arg-max decode), as one kernel (cnf_encoder_decode_actconv); returns (categories,
        running log-det)."""
        assert z.size(-1) == self.D, \
            "[!] ERROR in categorical decoding: Input must have %i latent dimensions but got %i" % (self.D, z.shape[-1])
        return ops.encoder_decode_actconv(z, act_bias, act_scales, conv_weight_inv, conv_sldj, self.class_table().detach(),
                                          self.category_prior, channel_padding_mask=channel_padding_mask, length=length, ldj=ldj)

    def _train_stats(self, z_out, class_prob_log, channel_padding_mask):
        """Monitoring scalars of the reference's train mode (:95-106); global reductions over the
        batch, not part of the likelihood — plain torch reductions on the device."""
        with torch.no_grad():
            pad = (channel_padding_mask.reshape(-1) if channel_padding_mask is not None
                   else torch.ones_like(class_prob_log))
            return {
                "avg_token_prob": (class_prob_log.exp() * pad).sum() / pad.sum(),
                "avg_token_bpd": -(class_prob_log * pad).sum() / pad.sum() * np.log2(np.exp(1)),
                "z_min": z_out.min(),
                "z_max": z_out.max(),
                "z_std": z_out.reshape(-1, z_out.shape[-1]).std(0).mean(),
            }

    # ---- linear flows: composition of layer kernels over [T*C,1,D] ---------------------------------
    def _forward_composed(self, z_categ, beta, channel_padding_mask, noise):
        B, N = z_categ.size(0), z_categ.size(1)
        T = B * N
        z_categ = z_categ.reshape(T, 1)
        pad = (channel_padding_mask.reshape(T, 1, -1) if channel_padding_mask is not None
               else z_categ.new_ones((T, 1, 1), dtype=torch.float32))
        z_cont = self._noise(T, z_categ.device, noise)
        init_log_p = self.prior_distribution.log_prob(z_cont).sum(dim=[1, 2])
        z_cont, ldj_forward = self._flow_forward(z_cont, z_categ, reverse=False)
        if not self.use_decoder:
            class_prior_log = torch.take(self.category_prior, z_categ.squeeze(dim=-1))
            log_point_prob = init_log_p - ldj_forward + class_prior_log
            class_prob_log = self._calculate_true_posterior(z_cont, z_categ, log_point_prob)
        else:
            class_prob_log = self._decoder_forward(z_cont, z_categ)
        ldj_loc = (beta * class_prob_log - (init_log_p - ldj_forward)) * pad.squeeze()
        z_cont = z_cont * pad
        detail = self._train_stats(z_cont, class_prob_log, pad) if self.training else {}
        return z_cont.reshape(B, N, -1), ldj_loc.reshape(B, N).sum(dim=-1), detail

    def _flow_forward(self, z_cont, z_categ, reverse, **kwargs):
        ldj = z_cont.new_zeros(z_cont.size(0), dtype=torch.float32)
        embed_features = self.embed_layer(z_categ)
        for flow in (self.flow_layers if not reverse else reversed(self.flow_layers)):
            z_cont, ldj = flow(z_cont, ldj, ext_input=embed_features, reverse=reverse, **kwargs)
        return z_cont, ldj

    def _decoder_forward(self, z_cont, z_categ, **kwargs):
        return self.decoder(z_cont).gather(dim=-1, index=z_categ.view(-1, 1))

    def _all_class_scores(self, z_cont):
        """reverse-flow log-prob + ldj + prior for every class: [T, C] (:155-164 / :186-195)."""
        T, C = z_cont.size(0), self.num_categories
        z_rep = z_cont.expand(-1, C, -1).reshape(-1, 1, z_cont.size(2))
        cls = torch.arange(C, dtype=torch.long, device=z_cont.device)[None, :].expand(T, -1).reshape(-1, 1)
        z_back, ldj_backward = self._flow_forward(z_rep, cls, reverse=True)
        back_log_p = self.prior_distribution.log_prob(z_back).sum(dim=[1, 2])
        return (back_log_p + ldj_backward).view(T, C) + self.category_prior[None, :]

    def _calculate_true_posterior(self, z_cont, z_categ, log_point_prob, **kwargs):
        scores = self._all_class_scores(z_cont)
        own = one_hot(z_categ.squeeze(), num_classes=scores.size(1))
        scores = scores * (1 - own) + log_point_prob.unsqueeze(dim=-1) * own      # forward value for the true class
        return log_point_prob - torch.logsumexp(scores, dim=-1)

    def _posterior_sample(self, z_cont, **kwargs):
        return self._all_class_scores(z_cont).argmax(dim=-1)

    def _decoder_sample(self, z_cont, **kwargs):
        return self.decoder(z_cont).argmax(dim=-1)

    def info(self):
        s = ""
        if len(self.flow_layers) > 1:
            s += "Linear Encodings of categories, with %i dimensions and %i flows.\n" % (self.D, len(self.flow_layers))
        else:
            s += "Mixture model encoding of categories with %i dimensions\n" % (self.D)
        s += "-> Prior distribution: %s\n" % self.prior_distribution.info()
        if self.use_decoder:
            s += "-> Decoder network: %s\n" % self.decoder.info()
        s += "\n".join(["-> [%i] " % (i + 1) + flow.info() for i, flow in enumerate(self.flow_layers)])
        return s


def _create_flows(num_dims, embed_dims, config):
    """linear_encoding.py:214-250 — [ExtActNorm] (mixture model) or n x [ExtActNorm, InvConv, Coupling]."""
    num_flows = get_param_val(config, "num_flows", 0)
    num_hidden_layers = get_param_val(config, "hidden_layers", 2)
    hidden_size = get_param_val(config, "hidden_size", 256)

    def actnorm():
        return ExtActNormFlow(c_in=num_dims, net=SimpleLinearLayer(c_in=embed_dims, c_out=2 * num_dims, data_init=True))

    def coupling():
        # affine (not mixture) couplings: the inverse must stay differentiable here
        return CouplingLayer(c_in=num_dims, mask=CouplingLayer.create_channel_mask(c_in=num_dims),
                             block_type="LinearNet",
                             model_func=lambda c_out: LinearNet(c_in=num_dims, c_out=c_out, num_layers=num_hidden_layers,
                                                                hidden_size=hidden_size, ext_input_dims=embed_dims))

    layers = []
    if num_flows == 0 or num_dims == 1:
        layers.append(actnorm())
    else:
        for _ in range(num_flows):
            layers += [actnorm(), InvertibleConv(c_in=num_dims), coupling()]
    return nn.ModuleList(layers)
