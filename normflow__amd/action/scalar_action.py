"""Scalar phi^4 lattice action (reference: src/action/scalar_action.py)."""
import torch

from .. import _hip


class ScalarPhi4Action:
    r"""S = sum_x [ kappa/2 (d_mu phi)^2 + m^2/2 phi^2 + lambda phi^4 ], lattice units
    of spacing `a` absorbed into the couplings (scalar_action.py:9-36)."""

    def __init__(self, *, m_sq, lambd, kappa=1, a=1):
        self.kappa, self.m_sq, self.lambd, self.a = kappa, m_sq, lambd, a

    def get_coef(self, lat_ndim):
        """(w0, w2, w4): hopping, quadratic and quartic weights."""
        kap = self.kappa * self.a ** (lat_ndim - 2)
        w0 = kap
        w2 = 0.5 * (self.m_sq * self.a ** lat_ndim + 2 * kap * lat_ndim)
        w4 = self.lambd * self.a ** lat_ndim
        return w0, w2, w4

    def __call__(self, cfgs):
        return self.action(cfgs)

    def action(self, cfgs):
        """Per-sample action of a batch of configurations (B, *L) -> (B,).  Device tensors take the
        one-pass HIP kernel (nf_phi4_action); host tensors the op chain below (the reference's)."""
        axes = tuple(range(1, cfgs.ndim))
        w0, w2, w4 = self.get_coef(cfgs.ndim - 1)
        if _hip.endpoint_supported(cfgs) and all(cfgs.shape[1:]):
            return _hip.Phi4ActionFn.apply(cfgs, float(w0), float(w2), float(w4))
        sq = cfgs * cfgs
        local = (w2 + w4 * sq) * sq
        hop = sum(cfgs * torch.roll(cfgs, 1, mu) for mu in axes) if axes else 0
        dens = local - w0 * hop
        return dens.sum(dim=axes) if axes else dens

    def action_density(self, cfgs):
        """A symmetric, kinetic-positive density whose sum is the action (:48-62)."""
        axes = tuple(range(1, cfgs.ndim))
        w0, w2, w4 = self.get_coef(cfgs.ndim - 1)
        w2 = w2 - w0 * (cfgs.ndim - 1)
        dens = w2 * cfgs ** 2 + w4 * cfgs ** 4
        for mu in axes:
            for step in (-1, 1):
                dens = dens + (w0 / 4) * (cfgs - torch.roll(cfgs, step, mu)) ** 2
        return dens

    def potential(self, x):
        return self.m_sq * x ** 2 + self.lambd * x ** 4

    def log_prob(self, x, action_logz=0):
        return -self.action(x) - action_logz
