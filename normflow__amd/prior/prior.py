"""Prior distributions feeding the flow (reference: src/prior/prior.py)."""
import copy
import math
from abc import ABC, abstractmethod

import torch

from .. import _hip


class Prior(ABC):
    """Wraps a torch.distributions object: sample(), sample_() -> (x, log r), log_prob()."""

    propagate_density = False

    def __init__(self, dist, seed=None):
        self.dist = dist
        Prior.manual_seed(seed)

    def sample(self, batch_size=1):
        return self.dist.sample((batch_size,))

    def sample_(self, batch_size=1):
        x = self.sample(batch_size)
        return x, self.log_prob(x)

    def log_prob(self, x):
        dens = self.dist.log_prob(x)
        if self.propagate_density or dens.dim() < 2:
            return dens
        return dens.sum(dim=tuple(range(1, dens.dim())))

    @staticmethod
    def manual_seed(seed):
        if isinstance(seed, int):
            torch.manual_seed(seed)

    @property
    def nvar(self):
        return math.prod(self.shape)

    @abstractmethod
    def to(self, *args, **kwargs):
        """Move the distribution's parameters (and hence its samples) to a device/dtype."""

    @property
    @abstractmethod
    def parameters(self):
        """dict of the tensors defining the prior."""


class UniformPrior(Prior):
    """Uniform on [low, high] (prior.py:65-91)."""

    def __init__(self, low=None, high=None, shape=None, seed=None, **kwargs):
        if shape is not None:
            low, high = torch.zeros(shape), torch.ones(shape)
        else:
            shape = low.shape
        super().__init__(torch.distributions.uniform.Uniform(low, high), seed, **kwargs)
        self.shape = shape

    def to(self, *args, **kwargs):
        self.dist.low = self.dist.low.to(*args, **kwargs)
        self.dist.high = self.dist.high.to(*args, **kwargs)

    @property
    def parameters(self):
        return dict(low=self.dist.low, high=self.dist.high)


class NormalPrior(Prior):
    """Normal(loc, scale); shape=... gives a unit normal on that lattice (prior.py:92-125).
    On the device the per-sample log-density is ONE fused pass (nf_normal_logprob)."""

    def log_prob(self, x):
        loc, scale = self.dist.loc, self.dist.scale
        if (not self.propagate_density and _hip.endpoint_supported(x) and x.dim() >= 2
                and tuple(x.shape[1:]) == tuple(loc.shape) and loc.device == x.device and loc.dtype == x.dtype):
            return _hip.NormalLogProbFn.apply(x, loc.contiguous(), scale.contiguous())
        return super().log_prob(x)

    def __init__(self, loc=None, scale=None, shape=None, seed=None, **kwargs):
        if shape is not None:
            loc, scale = torch.zeros(shape), torch.ones(shape)
        else:
            shape = loc.shape
        super().__init__(torch.distributions.normal.Normal(loc, scale), seed, **kwargs)
        self.shape = shape

    def setup_blockupdater(self, block_len):
        chopped = NormalPrior(loc=self.dist.loc.ravel()[:block_len], scale=self.dist.scale.ravel()[:block_len])
        self.blockupdater = BlockUpdater(chopped, block_len)

    def to(self, *args, **kwargs):
        self.dist.loc = self.dist.loc.to(*args, **kwargs)
        self.dist.scale = self.dist.scale.to(*args, **kwargs)

    @property
    def parameters(self):
        return dict(loc=self.dist.loc, scale=self.dist.scale)


class PriorList:
    """A list of priors sampled together (prior.py:128-157)."""

    def __init__(self, prior_list):
        self.prior_list = prior_list

    def sample(self, batch_size=1):
        return [p.sample(batch_size) for p in self.prior_list]

    def sample_(self, batch_size=1):
        x = self.sample(batch_size)
        return x, self.log_prob(x)

    def log_prob(self, x):
        return [p.log_prob(xi) for p, xi in zip(self.prior_list, x)]

    @property
    def nvar(self):
        return sum(p.nvar for p in self.prior_list)

    def to(self, *args, **kwargs):
        for p in self.prior_list:
            p.to(*args, **kwargs)

    @property
    def parameters(self):
        return [p.parameters for p in self.prior_list]


class BlockUpdater:
    """In-place refresh of one block of variables, with undo (prior.py:160-178)."""

    def __init__(self, chopped_prior, block_len):
        self.block_len = block_len
        self.chopped_prior = chopped_prior
        self.backup_block = None

    def _blocks(self, x):
        return x.view(x.shape[0], -1, self.block_len)

    def __call__(self, x, block_ind):
        view = self._blocks(x)
        self.backup_block = copy.deepcopy(view[:, block_ind])
        view[:, block_ind] = self.chopped_prior.sample(x.shape[0])

    def restore(self, x, block_ind, restore_ind=slice(None)):
        self._blocks(x)[restore_ind, block_ind] = self.backup_block[restore_ind]
