"""Lattice phi^4 action, the target density of the flow (API of src/action/scalar_action.py).

    S[phi] = sum_x ( w2 phi^2 + w4 phi^4 ) - w0 sum_mu sum_x phi(x) phi(x - mu)        (periodic)
with w0 = kappa a^(d-2), w2 = (m^2 a^d + 2 d kappa a^(d-2)) / 2, w4 = lambda a^d.
On the device the whole thing is one pass of the `nf_phi4_action` kernel (with its VJP).
"""
import torch

from .. import _hip


class ScalarPhi4Action:

    def __init__(self, *, m_sq, lambd, kappa=1, a=1):
        self.m_sq, self.lambd, self.kappa, self.a = m_sq, lambd, kappa, a

    def get_coef(self, lat_ndim):
        """(w0, w2, w4) for a `lat_ndim`-dimensional lattice."""
        hop = self.kappa * self.a ** (lat_ndim - 2)
        vol = self.a ** lat_ndim
        return hop, (self.m_sq * vol + 2 * lat_ndim * hop) / 2, self.lambd * vol

    def action(self, cfgs):
        """(B, *L) configurations -> (B,) actions."""
        d = cfgs.ndim - 1
        w0, w2, w4 = self.get_coef(d)
        if d >= 1 and cfgs.numel() and _hip.endpoint_supported(cfgs):
            return _hip.Phi4ActionFn.apply(cfgs, float(w0), float(w2), float(w4))
        # host tensors: site-local part, then one nearest-neighbour product per direction
        flat = lambda t: t.flatten(1).sum(dim=1) if d >= 1 else t
        sq = cfgs.square()
        total = flat(sq * (w2 + w4 * sq))
        for mu in range(1, d + 1):
            total = total - w0 * flat(cfgs * cfgs.roll(1, dims=mu))
        return total

    __call__ = action

    def potential(self, x):
        return self.m_sq * x ** 2 + self.lambd * x ** 4

    def log_prob(self, x, action_logz=0):
        """log density up to the constant `action_logz`."""
        return -self.action(x) - action_logz
