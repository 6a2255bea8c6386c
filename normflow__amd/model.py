"""`Model`: the user-facing bundle of prior, flow network and action, with sampling helpers.

API of the reference's src/_normflowcore.py (Model, Posterior, backward_sanitychecker); training
lives in `fitter.py`.
"""
import torch

from .device import ModelDeviceHandler
from .fitter import Fitter
from .mcmc import MCMCSampler


class Posterior:
    """Raw draws from the trained flow (no accept/reject)."""

    # MI355X-side option (no counterpart in the reference): push the draws through the net by replaying ONE HIP graph per batch
    # shape (graphs.GraphedFlow) -- for the launch-bound small lattices (16 x 16: 0.19 -> 0.08 ms per pass); same numbers,
    # re-captured by itself when the parameters change
    graphed = False

    def __init__(self, model):
        self._model = model
        self._graphs = {}

    def _flow(self, x):
        if not (self.graphed and x.is_cuda):
            return self._model.net_(x)
        key = (tuple(x.shape), x.dtype, x.device)
        flow = self._graphs.get(key)
        if flow is None:
            from .graphs import GraphedFlow
            if len(self._graphs) >= 4:             # (a graph keeps its activations: a few batch shapes at most)
                self._graphs.clear()
            flow = self._graphs[key] = GraphedFlow(self._model.net_, x)
        return flow(x)

    @torch.no_grad()
    def sample_(self, batch_size=1, preprocess_func=None):
        """(y, log q(y)): push prior draws through the net; log q = log r - log|J|."""
        m = self._model
        x, logr = m.prior.sample_(batch_size)
        if preprocess_func is not None:
            x, logr = preprocess_func(x, logr)
        y, logJ = self._flow(x)
        return y, logr - logJ

    @torch.no_grad()
    def sample(self, batch_size=1, **kwargs):
        return self.sample_(batch_size=batch_size, **kwargs)[0]

    @torch.no_grad()
    def sample__(self, batch_size=1, **kwargs):
        """(y, log q(y), log p(y)) with the unnormalised target log p = -S."""
        y, logq = self.sample_(batch_size=batch_size, **kwargs)
        return y, logq, -self._model.action(y)

    @torch.no_grad()
    def log_prob(self, y):
        """log q(y) by pulling y back through the inverse flow."""
        m = self._model
        x, logJ_inv = m.net_.backward(y)
        return m.prior.log_prob(x) + logJ_inv


class Model:
    """Model(prior=..., net_=..., action=...): `.fit(...)` trains, `.posterior.sample(n)` draws,
    `.mcmc.sample(n)` draws with Metropolis correction, `.device_handler` places / parallelises."""

    def __init__(self, *, prior, net_, action, name=None):
        self.prior, self.net_, self.action, self.name = prior, net_, action, name
        self.fit = Fitter(self)
        self.posterior = self.raw_dist = Posterior(self)
        self.mcmc = MCMCSampler(self)
        self.device_handler = ModelDeviceHandler(self)

    def transform(self, x):
        return self.net_(x)[0]


@torch.no_grad()
def backward_sanitychecker(model, n_samples=5, net_=None, return_details=False):
    """Print |x - f^-1(f(x))| and the residual log-Jacobian of a forward-backward round trip."""
    net_ = net_ or model.net_
    x = model.prior.sample(n_samples)
    y, logJ = net_(x)
    x_back, log_left = net_.backward(y, log0=logJ)
    print("Sanity check is OK if following numbers are zero up to round off:")
    print(f"{(x - x_back).abs().sum().item():g}", f"{log_left.abs().sum().item():g}")
    if return_details:
        return (x, y, x_back), (logJ, log_left)
