"""Priors: where the flow's input samples and their log-density come from.

API kept from the reference (src/prior/prior.py): `sample(B)`, `sample_(B) -> (x, log r)`,
`log_prob(x)`, `.to(...)`, `.shape`, `.nvar`, `.parameters`, `.dist`.  The implementation is this
package's own: a prior is described by two tensors (location / scale, or low / high).  On a HIP
device `NormalPrior.sample` and `sample_` BOTH draw through the fused Philox kernel
(`nf_normal_sample`: one launch emits x and log r), keyed by torch's CUDA generator: after the
same `torch.manual_seed`, `sample(B)` and `sample_(B)[0]` are the same configurations, as they
are in the reference (both go through `dist.sample`, prior.py:22-28).  The random STREAM is this
kernel's, not torch's (the reference's stream is whatever torch's sampler gives and is not part
of its contract); `NormalPrior(..., torch_rng=True)` keeps torch's sampler -- the reference's
stream on a given seed -- for both methods.  Elsewhere (CPU tensors, UniformPrior) samples are
drawn with the same torch generator calls `torch.distributions` makes.  The normal log-density
of device tensors is ONE fused HIP pass (`nf_normal_logprob`) instead of log_prob + sum.
"""
import math

import torch

from .. import _hip


def _batch_sum(t, keep):
    """Sum over everything but axis 0 unless densities are propagated."""
    return t if (keep or t.dim() < 2) else t.flatten(1).sum(dim=1)


class Prior:
    """Two-parameter prior over a lattice of `shape`; subclasses define draw / density."""

    propagate_density = False
    _names = ("a", "b")

    def __init__(self, first, second, seed=None):
        self._p = [first, second]
        self.shape = tuple(first.shape)
        self.manual_seed(seed)

    # -- protocol
    def sample(self, batch_size=1):
        return self._draw((batch_size,) + self.shape)

    def sample_(self, batch_size=1):
        x = self.sample(batch_size)
        return x, self.log_prob(x)

    def log_prob(self, x):
        return _batch_sum(self._density(x), self.propagate_density)

    @staticmethod
    def manual_seed(seed):
        if isinstance(seed, int):
            torch.manual_seed(seed)

    @property
    def nvar(self):
        return math.prod(self.shape)

    def to(self, *args, **kwargs):
        """Move the defining tensors (and hence future samples) to a device / dtype."""
        self._p = [t.to(*args, **kwargs) for t in self._p]

    @property
    def parameters(self):
        return dict(zip(self._names, self._p))

    def _draw(self, full_shape):
        raise NotImplementedError

    def _density(self, x):
        raise NotImplementedError


class NormalPrior(Prior):
    """Independent normals; `shape=L` means zero mean, unit width on that lattice."""

    _names = ("loc", "scale")

    def __init__(self, loc=None, scale=None, shape=None, seed=None, torch_rng=False):
        """`torch_rng=True` keeps torch.distributions' sampler (the reference's random stream on a given seed) and the
        separate log_prob pass; the default on a HIP device is the fused Philox kernel: one launch emits x and log r."""
        self._unit = shape is not None
        if shape is not None:
            loc, scale = torch.zeros(shape), torch.ones(shape)
        super().__init__(loc, scale, seed)
        self.torch_rng = torch_rng

    def _kernel_draws(self, batch_size):
        loc = self.loc
        return (not self.torch_rng and loc.is_cuda and loc.dtype in (torch.float32, torch.float64) and batch_size >= 1)

    def _kernel_sample(self, batch_size):
        unit = self._unit
        return _hip.normal_sample(None if unit else self.loc.reshape(-1), None if unit else self.scale.reshape(-1),
                                  batch_size, self.shape, self.loc.dtype, self.loc.device)

    def sample(self, batch_size=1):
        """x alone (prior.py:22-24), from the same generator and kernel as `sample_`: one seed, one stream."""
        if self._kernel_draws(batch_size):
            return self._kernel_sample(batch_size)[0]
        return super().sample(batch_size)

    def sample_(self, batch_size=1):
        """(x, log r) (prior.py:26-29): on a HIP device ONE kernel (nf_normal_sample) draws the field and accumulates its
        log-density from the normals still in registers."""
        if self._kernel_draws(batch_size):
            x, logr = self._kernel_sample(batch_size)
            return x, (self.log_prob(x) if self.propagate_density else logr)
        return super().sample_(batch_size)

    loc = property(lambda self: self._p[0])
    scale = property(lambda self: self._p[1])

    @property
    def dist(self):
        """The equivalent torch.distributions object (reference attribute)."""
        return torch.distributions.normal.Normal(self.loc, self.scale)

    def _draw(self, full_shape):
        with torch.no_grad():      # what Normal.sample does: one torch.normal on expanded parameters
            return torch.normal(self.loc.expand(full_shape), self.scale.expand(full_shape))

    def _density(self, x):
        z = (x - self.loc) / self.scale
        return -0.5 * z * z - torch.log(self.scale) - 0.5 * math.log(2 * math.pi)

    def log_prob(self, x):
        fused = (not self.propagate_density and x.dim() >= 2 and _hip.endpoint_supported(x)
                 and tuple(x.shape[1:]) == self.shape and self.loc.device == x.device
                 and self.loc.dtype == x.dtype)
        if fused:
            return _hip.NormalLogProbFn.apply(x, self.loc.contiguous(), self.scale.contiguous())
        return super().log_prob(x)


class UniformPrior(Prior):
    """Independent uniforms on [low, high); `shape=L` means the unit interval."""

    _names = ("low", "high")

    def __init__(self, low=None, high=None, shape=None, seed=None):
        if shape is not None:
            low, high = torch.zeros(shape), torch.ones(shape)
        super().__init__(low, high, seed)

    low = property(lambda self: self._p[0])
    high = property(lambda self: self._p[1])

    @property
    def dist(self):
        return torch.distributions.uniform.Uniform(self.low, self.high)

    def _draw(self, full_shape):
        with torch.no_grad():
            u = torch.rand(full_shape, dtype=self.low.dtype, device=self.low.device)
            return self.low + u * (self.high - self.low)

    def _density(self, x):
        inside = (x >= self.low) & (x < self.high)
        return torch.where(inside, -torch.log(self.high - self.low), torch.full_like(x, -math.inf))
