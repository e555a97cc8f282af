"""Ordered container of flow layers with the running log-det sum.

Interface of layers/flows/flow_model.py (FlowModel.forward :25-53, run_data_init_layer :104-131).
Differences that are deliberate: the per-layer `torch.isnan(z).sum() == 0` host sync (:42) is replaced
by device-side flag bits that the kernels raise and that are checked ONCE at the end of the pass
(same AssertionError), unless CNF_STRICT_ASSERTS=1 asks for the per-layer check."""
import torch
import torch.nn as nn

from ... import functional as Fn
from ... import ops
from .activation_normalization import ActNormFlow
from .permutation_layers import InvertibleConv


class FlowModel(nn.Module):

    def __init__(self, layers=None, name="Flow model"):
        super().__init__()
        self.flow_layers = nn.ModuleList()
        self.name = name
        if layers is not None:
            self.add_layers(layers)

    def add_layers(self, layers):
        for layer in layers:
            self.flow_layers.append(layer)
        self.print_overview()

    def forward(self, z, ldj=None, reverse=False, get_ldj_per_layer=False, **kwargs):
        nll_request = kwargs.pop("_nll", None)           # set by nll(): assemble the NLL, fused into the last layer if possible
        if ldj is None:
            ldj = z.new_zeros(z.size(0), dtype=torch.float32)
        order = list(enumerate(self.flow_layers))
        if reverse:
            order.reverse()
        per_layer = []
        fusable = self._fusable(z, get_ldj_per_layer)
        # the same fusions on the TRAINING path (forward direction, autograd on): one autograd.Function per fused group whose
        # backward recomputes the intermediate the forward kept in registers (functional.ActConvFn / MixtureActConvFn /
        # EncoderActConvFn)
        trainable = self._fusable_training(z, get_ldj_per_layer, reverse)
        skip = set()
        for pos, (index, layer) in enumerate(order):
            if index in skip:
                continue
            # the fusions of latents-in, latents-out layers need float32 latents AT THIS POINT of the pass (a flow that starts
            # with the categorical encoder is handed int64 categories: until round 3 that switched every fusion of the
            # encoding direction off, because the test looked at the pass's input only)
            fuse = fusable and z.dtype == torch.float32
            fuse_train = trainable and z.dtype == torch.float32
            if ((fuse or fuse_train) and not reverse and pos + 2 < len(order) and type(layer).__name__ == "MixtureCDFCoupling"
                    and type(order[pos + 1][1]) is ActNormFlow and type(order[pos + 2][1]) is InvertibleConv
                    and layer.c_in in ops.FUSED_ACTCONV_DIMS):
                # mixture coupling of this flow step + ActNorm + 1x1 conv of the next one in ONE kernel: the coupling's
                # output never goes to HBM in between (same arithmetic as the three layers, bit for bit)
                act, conv = order[pos + 1][1], order[pos + 2][1]
                pad = kwargs.get("channel_padding_mask", None)
                net_kwargs = {k: v for k, v in kwargs.items() if k != "channel_padding_mask"}
                nn_out = layer.run_network(x=z * layer._prepare_mask(layer.mask, z), **net_kwargs)
                weight, sldj = conv._get_weight(device_name=str(z.device), inverse=False)
                if fuse_train:
                    z, ldj = Fn.MixtureActConvFn.apply(
                        z, nn_out, layer.scaling_factor, layer.mixture_scaling_factor, act.bias, act.scales, weight, sldj, ldj,
                        layer.mask, pad, kwargs.get("length", None), layer.num_mixtures, layer.regularizer_max,
                        layer.regularizer_factor, layer.training)
                    skip.update((order[pos + 1][0], order[pos + 2][0]))
                    continue
                z, ldj, _ = ops.mixture_coupling_actconv(
                    z, nn_out, layer.mask, layer.num_mixtures, act.bias, act.scales, weight, sldj,
                    scaling_factor=layer.scaling_factor, mixture_scaling_factor=layer.mixture_scaling_factor,
                    channel_padding_mask=pad, length=kwargs.get("length", None), reg_max=layer.regularizer_max,
                    reg_factor=layer.regularizer_factor, is_training=layer.training, ldj=ldj, want_reg=False)
                skip.update((order[pos + 1][0], order[pos + 2][0]))
                continue
            if (fuse and pos + 2 < len(order) and type(layer).__name__ == "CouplingLayer" and layer.c_in in ops.FUSED_ACTCONV_DIMS
                    and type(order[pos + 1][1]) is (InvertibleConv if reverse else ActNormFlow)
                    and type(order[pos + 2][1]) is (ActNormFlow if reverse else InvertibleConv)):
                # affine coupling of this flow step + ActNorm + 1x1 conv of the next one (forward), or the coupling's inverse + the
                # inverted conv + ActNorm of its own step (reverse: the order the layers are walked in) in ONE kernel, the bits of
                # the three layers (cnf_affine_coupling_actconv; shapes outside that kernel run as coupling + fused pair)
                act, conv = (order[pos + 2][1], order[pos + 1][1]) if reverse else (order[pos + 1][1], order[pos + 2][1])
                pad = kwargs.get("channel_padding_mask", None)
                net_kwargs = {k: v for k, v in kwargs.items() if k != "channel_padding_mask"}
                nn_out = layer.run_network(x=z * layer._prepare_mask(layer.mask, z), **net_kwargs)
                weight, sldj = conv._get_weight(device_name=str(z.device), inverse=reverse)
                z, ldj = ops.affine_coupling_actconv(z, nn_out, layer.scaling_factor, layer.mask, act.bias, act.scales, weight, sldj,
                                                     reverse=reverse, length=kwargs.get("length", None), channel_padding_mask=pad, ldj=ldj)
                skip.update((order[pos + 1][0], order[pos + 2][0]))
                continue
            if ((fusable or trainable) and not reverse and pos + 2 < len(order) and type(layer).__name__ == "LinearCategoricalEncoding"
                    and not z.is_floating_point()
                    and type(order[pos + 1][1]) is ActNormFlow and type(order[pos + 2][1]) is InvertibleConv
                    and layer.fusable_with_actconv(differentiable=trainable) and order[pos + 1][1].c_in in ops.FUSED_ACTCONV_DIMS
                    and not isinstance(kwargs.get("beta", 1), torch.Tensor)):
                # encoder + ActNorm + 1x1 conv of the first flow step in ONE kernel: the latents go to HBM once, already
                # transformed (same arithmetic as the three layers, bit for bit)
                act, conv = order[pos + 1][1], order[pos + 2][1]
                weight, sldj = conv._get_weight(device_name=str(z.device), inverse=False)
                z, ldj = layer.forward_with_actconv(z, act.bias, act.scales, weight, sldj, ldj=ldj, beta=kwargs.get("beta", 1),
                                                    channel_padding_mask=kwargs.get("channel_padding_mask", None),
                                                    length=kwargs.get("length", None), noise=kwargs.get("noise", None),
                                                    differentiable=trainable)
                skip.update((order[pos + 1][0], order[pos + 2][0]))
                continue
            if (fuse and reverse and pos + 2 < len(order) and type(layer) is InvertibleConv and type(order[pos + 1][1]) is ActNormFlow
                    and type(order[pos + 2][1]).__name__ == "LinearCategoricalEncoding" and order[pos + 2][1]._is_mixture_model()
                    and order[pos + 1][1].c_in in ops.FUSED_ACTCONV_DIMS):
                # sampling direction: inverse 1x1 conv + inverse ActNorm of the first flow step + the arg-max decode in ONE
                # kernel (same arithmetic as the three layers, bit for bit)
                act, enc = order[pos + 1][1], order[pos + 2][1]
                weight, sldj = layer._get_weight(device_name=str(z.device), inverse=True)
                z, ldj = enc.decode_with_actconv(z, act.bias, act.scales, weight, sldj, ldj=ldj,
                                                 channel_padding_mask=kwargs.get("channel_padding_mask", None),
                                                 length=kwargs.get("length", None))
                skip.update((order[pos + 1][0], order[pos + 2][0]))
                continue
            if (fuse or fuse_train) and pos + 1 < len(order):
                pair = (layer, order[pos + 1][1]) if not reverse else (order[pos + 1][1], layer)
                if type(pair[0]) is ActNormFlow and type(pair[1]) is InvertibleConv and pair[0].c_in in ops.FUSED_ACTCONV_DIMS:
                    # ActNorm -> 1x1 conv (or the pair backwards) in one kernel; same arithmetic as the two layers
                    weight, sldj = pair[1]._get_weight(device_name=str(z.device), inverse=reverse)
                    if fuse_train:
                        z, ldj = Fn.ActConvFn.apply(z, pair[0].bias, pair[0].scales, weight, sldj, ldj, kwargs.get("length", None),
                                                    kwargs.get("channel_padding_mask", None))
                        skip.add(order[pos + 1][0])
                        continue
                    z, ldj = ops.actnorm_invconv(z, pair[0].bias, pair[0].scales, weight, sldj, reverse=reverse,
                                                 length=kwargs.get("length", None),
                                                 channel_padding_mask=kwargs.get("channel_padding_mask", None), ldj=ldj)
                    skip.add(order[pos + 1][0])
                    continue
            # the differentiable NLL Functions are built for the unit logistic prior (their backward kernels hard-code it): a
            # request with another sigma — e.g. one made under no_grad and used with grad on — must not take them
            unit_prior = nll_request is not None and float(nll_request["sigma"]) == ops.LOGISTIC_SIGMA
            last_with_nll = (nll_request is not None and pos == len(order) - 1 and not reverse and z.is_cuda and z.dtype == torch.float32
                             and (not torch.is_grad_enabled() or (trainable and unit_prior)))
            if last_with_nll and type(layer).__name__ == "CouplingLayer":
                # last layer = affine coupling: transform + prior log-prob + NLL in one kernel
                pad = kwargs.get("channel_padding_mask", None)
                net_kwargs = {k: v for k, v in kwargs.items() if k != "channel_padding_mask"}
                nn_out = layer.run_network(x=z * layer._prepare_mask(layer.mask, z), **net_kwargs)
                if torch.is_grad_enabled():
                    nll_request["nll"], z, ldj = Fn.AffineCouplingNllFn.apply(
                        z, nn_out, layer.scaling_factor, ldj, layer.mask, pad, nll_request["length"],
                        nll_request["sigma"], nll_request["log_sigma"])
                    if nll_request["sums"] is not None:
                        ops.nll_sum(nll_request["nll"].detach(), nll_request["sums"])
                    continue
                z, ldj, _, nll_request["nll"] = ops.affine_coupling_nll(
                    z, nn_out, layer.scaling_factor, layer.mask, ldj=ldj, length=nll_request["length"],
                    channel_padding_mask=pad, sums=nll_request["sums"],
                    sigma=nll_request["sigma"], log_sigma=nll_request["log_sigma"])
                continue
            if last_with_nll and type(layer).__name__ == "MixtureCDFCoupling":
                # last layer = mixture-CDF coupling (every flow of the four experiments ends in one, e.g.
                # experiments/set_modeling/flow_model.py:58-62): transform + prior log-prob + NLL in one kernel
                pad = kwargs.get("channel_padding_mask", None)
                net_kwargs = {k: v for k, v in kwargs.items() if k != "channel_padding_mask"}
                nn_out = layer.run_network(x=z * layer._prepare_mask(layer.mask, z), **net_kwargs)
                if torch.is_grad_enabled():
                    nll_request["nll"], z, ldj = Fn.MixtureCouplingNllFn.apply(
                        z, nn_out, layer.scaling_factor, layer.mixture_scaling_factor, ldj, layer.mask, pad, nll_request["length"],
                        layer.num_mixtures, layer.regularizer_max, layer.regularizer_factor, layer.training,
                        nll_request["sigma"], nll_request["log_sigma"])
                else:
                    z, ldj, reg, _, nll_request["nll"] = ops.mixture_coupling_nll(
                        z, nn_out, layer.mask, layer.num_mixtures, layer.scaling_factor, layer.mixture_scaling_factor,
                        channel_padding_mask=pad, reg_max=layer.regularizer_max, reg_factor=layer.regularizer_factor,
                        is_training=layer.training, ldj=ldj, length=nll_request["length"],
                        sigma=nll_request["sigma"], log_sigma=nll_request["log_sigma"])
                if nll_request["sums"] is not None:
                    ops.nll_sum(nll_request["nll"].detach(), nll_request["sums"])
                continue
            res = layer(z, reverse=reverse, get_ldj_per_layer=get_ldj_per_layer, **kwargs)
            if len(res) == 2:
                z, layer_ldj = res
                detail = layer_ldj
            elif len(res) == 3:
                z, layer_ldj, detail = res
            else:
                print("[!] ERROR: Got more return values than expected: %i" % (len(res)))
            if ops._STRICT and z.is_cuda:
                ops.check_flags(z.device, "Layer (%i):\n%s" % (index + 1, layer.info()))
            ldj = ldj + layer_ldj
            if isinstance(detail, list):
                per_layer += detail
            else:
                per_layer.append(detail)
        if nll_request is not None and "nll" not in nll_request:
            if torch.is_grad_enabled() and (z.requires_grad or ldj.requires_grad):
                if not z.is_cuda or float(nll_request["sigma"]) != ops.LOGISTIC_SIGMA:
                    raise NotImplementedError("FlowModel: the differentiable NLL (PriorNllFn) runs on the device under the unit logistic "
                                              "prior only (sigma %s, device %s): assemble the NLL from z and ldj instead"
                                              % (nll_request["sigma"], z.device))
                nll_request["nll"] = Fn.PriorNllFn.apply(z, ldj, nll_request["length"], kwargs.get("channel_padding_mask", None))
                if nll_request["sums"] is not None:
                    ops.nll_sum(nll_request["nll"].detach(), nll_request["sums"])
            else:
                _, nll_request["nll"] = ops.prior_nll(z, ldj, nll_request["length"], kwargs.get("channel_padding_mask", None),
                                                      sums=nll_request["sums"], sigma=nll_request["sigma"],
                                                      log_sigma=nll_request["log_sigma"])
        if z.is_cuda:
            ops.check_flags(z.device, "Flow: %s" % self.name)
        if nll_request is not None:
            # the NLL is RETURNED as well (a wrapper such as DistributedDataParallel hands the module a copy of its keyword
            # containers: the caller's dict would stay empty)
            return (z, ldj, nll_request["nll"], per_layer) if get_ldj_per_layer else (z, ldj, nll_request["nll"])
        if get_ldj_per_layer:
            return z, ldj, per_layer
        return z, ldj

    def nll_request(self, length=None, prior=None, sums=None):
        """The `_nll` keyword of `forward`: ask the pass to assemble the per-sample NLL under a zero-mean logistic prior as well —
        fused into the last coupling layer where there is one — and leave it in the returned dict's "nll".  With autograd on
        the result is differentiable (functional.MixtureCouplingNllFn / AffineCouplingNllFn / PriorNllFn), so a training loop
        can go through a wrapper module's forward (DistributedDataParallel); a pass given `_nll` returns (z, ldj, nll):
            z, ldj, nll = ddp(x, length=ln, _nll=flow.nll_request(length=ln)); loss = nll.mean()"""
        from .distributions import LogisticDistribution
        prior = prior if prior is not None else LogisticDistribution()
        if type(prior) is not LogisticDistribution or float(prior.mu) != 0.0:
            raise NotImplementedError("FlowModel.nll: zero-mean LogisticDistribution prior only")
        if float(prior.sigma) != ops.LOGISTIC_SIGMA and torch.is_grad_enabled():
            raise NotImplementedError("the differentiable NLL assembly is built for the unit logistic prior")
        return {"sigma": float(prior.sigma), "log_sigma": float(prior.log_sigma), "sums": sums, "length": length}

    @torch.no_grad()
    def nll(self, z, length=None, prior=None, sums=None, **kwargs):
        """Evaluation shortcut: per-sample negative log-likelihood per element of `z` under the flow and a logistic
        prior, i.e. `forward` followed by the NLL assembly of the task classes (experiments/set_modeling/task.py:96-118:
        nll_b = (-ldj_b - sum_{n,d} log p(z_bnd) * pad_bn) / length_b).  Returns (z, ldj, nll [B]); `sums` (fp64 [2],
        optional) receives (sum_b nll_b, B) — the pair that is all-reduced over ranks.

        When the last flow layer is an affine `CouplingLayer` or a `MixtureCDFCoupling`, that layer and the NLL
        assembly run as ONE kernel (cnf_affine_coupling_nll / cnf_mixture_coupling_nll: the prior term is accumulated
        while z is still in registers); otherwise the separate prior kernel is used.  Same numbers either way (tests).  Goes through `self.forward`, so the masks a
        subclass builds from `length` reach the layers as usual.  `nll_loss` is the differentiable twin."""
        request = self.nll_request(length, prior, sums)
        if length is not None:
            kwargs["length"] = length
        return self.forward(z, reverse=False, _nll=request, **kwargs)

    def nll_loss(self, z, length=None, prior=None, **kwargs):
        """`nll` with autograd on: (z, ldj, nll [B]) with a differentiable nll (training loops: `nll.mean().backward()`)."""
        request = self.nll_request(length, prior, None)
        if length is not None:
            kwargs["length"] = length
        return self.forward(z, reverse=False, _nll=request, **kwargs)

    def _fusable(self, z, get_ldj_per_layer):
        """Layer fusion only where it is unobservable: no autograd, no per-layer log-det report, CUDA tensors."""
        return (not get_ldj_per_layer and not torch.is_grad_enabled() and isinstance(z, torch.Tensor) and z.is_cuda
                and ops.FUSE_LAYERS)

    def _fusable_training(self, z, get_ldj_per_layer, reverse):
        """Layer fusion with autograd on: forward direction, no per-layer report (the fused groups have no per-layer log-det and
        the encoder's monitoring scalars are not computed), CUDA tensors; ops.FUSE_TRAINING = False restores one Function per layer."""
        return (not reverse and not get_ldj_per_layer and torch.is_grad_enabled() and isinstance(z, torch.Tensor) and z.is_cuda
                and ops.FUSE_LAYERS and ops.FUSE_TRAINING)

    def reverse(self, z):
        """The inverse pass.  (The reference's one-liner, flow_model.py:56-57, passes an undefined name and can only
        raise NameError; nothing calls it.)"""
        return self.forward(z, reverse=True)

    def test_reversibility(self, z, **kwargs):
        """flow_model.py:60-78 — per-layer fwd∘inv check, exact-zero criterion like the reference."""
        failed = False
        for index, layer in enumerate(self.flow_layers):
            z_layer, ldj_layer = layer(z, reverse=False, **kwargs)[:2]
            z_rec, ldj_rec = layer(z_layer, reverse=True, **kwargs)[:2]
            if (z_layer - z_rec).abs().sum() != 0 or (ldj_layer + ldj_rec).abs().sum() != 0:
                print("-" * 100)
                print("[!] WARNING: Reversibility check failed for layer index %i" % index)
                print(layer.info())
                print("-" * 100)
                failed = True
        print("+" * 100)
        print("Reversibility test %s (tested %i layers)" % ("failed" if failed else "succeeded", len(self.flow_layers)))
        print("+" * 100)

    def get_inner_activations(self, z, reverse=False, return_names=False, **kwargs):
        outs, names = [z.detach()], []
        for layer in (self.flow_layers if not reverse else reversed(self.flow_layers)):
            z = layer(z, reverse=reverse, **kwargs)[0]
            outs.append(z.detach())
            names.append(layer.__class__.__name__)
        return (outs, names) if return_names else outs

    def initialize_data_dependent(self, batch_list):
        """batch_list: [(z, kwargs), ...] (flow_model.py:96-101)."""
        with torch.no_grad():
            for index, layer in enumerate(self.flow_layers):
                print("Processing layer %i..." % (index + 1), end="\r")
                batch_list = FlowModel.run_data_init_layer(batch_list, layer)

    @staticmethod
    def run_data_init_layer(batch_list, layer):
        """Initialise `layer` on the concatenation of all batches, then push every batch through it."""
        multi = isinstance(batch_list[0][0], (tuple, list))
        if layer.need_data_init():
            stacked = {}
            for key in batch_list[0][1].keys():
                vals = [b[1][key] for b in batch_list]
                stacked[key] = torch.cat(vals, dim=0) if isinstance(vals[0], torch.Tensor) else vals[0]
            if not multi:
                layer.data_init_forward(torch.cat([z for z, _ in batch_list], dim=0), **stacked)
            else:
                joined = [torch.cat([z[i] for z, _ in batch_list], dim=0) for i in range(len(batch_list[0][0]))]
                layer.data_init_forward(*joined, **stacked)
        outs = []
        for z, kwargs in batch_list:
            if multi:
                res = layer(*z, reverse=False, **kwargs)
                outs.append([e.detach() for e in res[:-1] if isinstance(e, torch.Tensor)])
                if len(res) == 4 and isinstance(res[-1], dict):
                    kwargs.update(res[-1])
                    outs[-1] = outs[-1][:-1]
            else:
                outs.append(layer(z, reverse=False, **kwargs)[0].detach())
        return [(outs[i], batch_list[i][1]) for i in range(len(batch_list))]

    def need_data_init(self):
        return any(flow.need_data_init() for flow in self.flow_layers)

    def print_overview(self):
        lines = ["(%2i) %s" % (i + 1, layer.info()) for i, layer in enumerate(self.flow_layers)]
        width = max([20] + [len(s) for s in "\n".join(lines).split("\n")])
        print("=" * width)
        print("%s with %i flows" % (self.name, len(self.flow_layers)))
        print("-" * width)
        print("\n".join(lines))
        print("=" * width)
