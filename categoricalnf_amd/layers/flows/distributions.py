"""Prior distributions over the latents (layers/flows/distributions.py).

LogisticDistribution (:91-185; sigma = 1/1.81) evaluates log_prob and the uniform->logistic map on
the HIP kernels.  Noise is drawn with PyTorch's generators (plumbing): on the CPU generator like
the reference when `device` is None (bit-compatible seeding), on the device generator otherwise."""
import sys

import numpy as np
import torch
import torch.nn as nn

from ... import compat
from ... import functional as Fn
from ... import ops
from ...host_utils import get_param_val


class PriorDistribution(nn.Module):

    GAUSSIAN = 0
    LOGISTIC = 1

    def __init__(self, **kwargs):
        super().__init__()
        self.distribution = self._create_distribution(**kwargs)

    def _create_distribution(self, **kwargs):
        raise NotImplementedError

    def forward(self, shape=None):
        return self.sample(shape=shape)

    def sample(self, shape=None):
        return self.distribution.sample() if shape is None else self.distribution.sample(sample_shape=shape)

    def log_prob(self, x):
        logp = self.distribution.log_prob(x)
        assert torch.isnan(logp).sum() == 0, "[!] ERROR: Found NaN values in log-prob of distribution."
        return logp

    def prob(self, x):
        return self.log_prob(x).exp()

    def icdf(self, x):
        assert ((x < 0) | (x > 1)).sum() == 0, \
            "[!] ERROR: Found values outside the range of 0 to 1 as input to the inverse cumulative distribution function."
        return self.distribution.icdf(x)

    def cdf(self, x):
        return self.distribution.cdf(x)

    def info(self):
        raise NotImplementedError

    @staticmethod
    def get_string_of_distributions():
        return "%i - Gaussian, %i - Logistic" % (PriorDistribution.GAUSSIAN, PriorDistribution.LOGISTIC)


class GaussianDistribution(PriorDistribution):
    """Normal prior (distributions.py:74-88): plain torch.distributions host code, off the HIP path (no experiment of the
    reference selects it; the logistic prior is the default everywhere).  `sample` takes the keyword arguments this
    package's callers pass to a prior (device, temp, return_ldj) so that it can stand in for the logistic one."""

    def __init__(self, mu=0.0, sigma=1.0, **kwargs):
        super().__init__(mu=mu, sigma=sigma)
        self.mu = mu
        self.sigma = sigma

    def _create_distribution(self, mu=0.0, sigma=1.0, **kwargs):
        return torch.distributions.normal.Normal(loc=mu, scale=sigma)

    def sample(self, shape=None, return_ldj=False, temp=1.0, device=None, **kwargs):
        x = super().sample(shape=shape)
        if temp != 1.0:
            x = self.mu + (x - self.mu) * temp
        if device is not None:
            x = x.to(device)
        return (x, -self.log_prob(x)) if return_ldj else x

    def info(self):
        return "Gaussian distribution with mu=%f and sigma=%f" % (self.mu, self.sigma)


class LogisticDistribution(PriorDistribution):

    def __init__(self, mu=0.0, sigma=1.0, eps=1e-4, **kwargs):
        sigma = sigma / 1.81       # std of a unit logistic is ~1.81 (:95)
        super().__init__(mu=mu, sigma=sigma)
        self.mu = mu
        self.sigma = sigma
        self.log_sigma = np.log(self.sigma)
        self.eps = eps

    def _create_distribution(self, mu=0.0, sigma=1.0, **kwargs):
        return torch.distributions.uniform.Uniform(low=0.0, high=1.0)

    def uniform(self, shape, generator="cpu", device=None):
        """U[0,1) draw of `shape`: on the global CPU generator like the reference (:139-140, so that
        torch.manual_seed reproduces the reference's noise) or on the device generator."""
        if generator == "cpu":
            return self.distribution.sample(sample_shape=shape)
        return torch.rand(tuple(shape), dtype=torch.float32, device=device)

    def sample(self, shape=None, return_ldj=False, temp=1.0, device=None, uniform=None, generator="cpu"):
        """Logistic sample on the GPU (`device`, default: current CUDA device).

        The uniform draw comes from `uniform` (injected, parity tests), the CPU generator
        (default, bit-compatible with the reference's seeding) or the device generator
        (generator="device", no H2D copy).  The squeeze / logit(fp64) / scale steps
        (:143-145,117-127) run in cnf_logistic_from_uniform."""
        if temp != 1.0:
            raise NotImplementedError("temperature sampling is broken in the reference itself (distributions.py:147)")
        if device is None:
            if not torch.cuda.is_available():
                raise ops.HipOnlyError("LogisticDistribution.sample needs a CUDA(HIP) device; there is no CPU path")
            device = torch.device("cuda", torch.cuda.current_device())
        u = uniform if uniform is not None else self.uniform(shape, generator=generator, device=device)
        if shape is not None and shape[-1] != 1 and u.dim() == len(shape) + 1:
            u = u.squeeze(dim=-1)
        u = u.to(device=device, dtype=torch.float32)
        x = ops.logistic_from_uniform(u, mu=self.mu, sigma=self.sigma, eps=self.eps)
        if not return_ldj:
            return x
        return x, -self.log_prob(x)

    def log_prob(self, x):
        if Fn.needs_grad(x):
            return Fn.LogisticLogProbFn.apply(x, self.mu, self.sigma, float(self.log_sigma))
        return ops.logistic_log_prob(x, mu=self.mu, sigma=self.sigma, log_sigma=float(self.log_sigma))

    def icdf(self, x, return_ldj=False):
        """Logistic quantile function z = mu + sigma logit(x) (distributions.py:166-173, shift_x :117-127): the
        uniform->logistic kernel without its eps squeeze; the log-det -log x - log(1-x) - log sigma is formed in fp64
        like the reference.  Not on the flows' path (no layer calls it); kept for API parity."""
        assert ((x < 0) | (x > 1)).sum() == 0, \
            "[!] ERROR: Found values outside the range of 0 to 1 as input to the inverse cumulative distribution function."
        z = ops.logistic_from_uniform(x.to(torch.float32), mu=self.mu, sigma=self.sigma, eps=0.0)
        if not return_ldj:
            return z
        xd = x.double()
        return z, (-torch.log(xd) - torch.log(1.0 - xd)).float() - float(self.log_sigma)

    def cdf(self, x, return_ldj=False):
        """Logistic CDF sigmoid((x - mu) / sigma) (distributions.py:176-181, unshift_x :129-136); its log-det is
        -log_prob(x), from the HIP log-prob kernel."""
        z = torch.sigmoid((x - self.mu) / self.sigma)
        if not return_ldj:
            return z
        return z, -self.log_prob(x)

    def info(self):
        return "Sigmoid Uniform distribution with mu=%.2f and sigma=%.2f" % (self.mu, self.sigma)


def create_prior_distribution(distribution_params):
    """distributions.py:190-200.  The logistic prior (the default of every experiment) runs on the HIP kernels; the
    optional Gaussian prior is plain torch.distributions host code."""
    kind = get_param_val(distribution_params, "distribution_type", PriorDistribution.LOGISTIC)
    params = {k: v for k, v in distribution_params.items() if v is not None}
    if kind == PriorDistribution.GAUSSIAN:
        return GaussianDistribution(**params)
    if kind == PriorDistribution.LOGISTIC:
        return LogisticDistribution(**params)
    print("[!] ERROR: Unknown distribution type %s" % str(kind))
    sys.exit(1)


# the reference's command-line flag builders (distributions.py:203-222) are host-side code of its training template:
# served from the user's checkout when it is on sys.path, by NAME — any other missing attribute is a plain AttributeError
_FROM_REFERENCE = ("add_prior_distribution_parameters", "prior_distribution_args_to_params")


def __getattr__(name):
    if name in _FROM_REFERENCE:
        return compat.fall_through("layers.flows.distributions", name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
