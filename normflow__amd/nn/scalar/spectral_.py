"""Spectral first block of the reference's example network: mean field + power spectrum.

    PSDBlock_(mfnet_, fftnet_):  x  ->  mfnet_(mean(x))  +  fftnet_(x - mean(x))

Mirrors src/nn/scalar/psd_.py:17-57, src/nn/scalar/meanfield_.py:17-66 and
src/nn/scalar/fftflow_.py:37-349 (FFTNet_, IPSD, FreeScalar lattice k^2), so that
examples/scalar_affine.py assembles its network from this package unchanged.

Where the work runs: the FFTs are torch.fft (rocFFT) -- plumbing, as SURVEY section 8(f) says --
and the learned curves are the package's own HIP paths: the inverse power spectrum is ONE shared
rational-quadratic spline evaluated on the k^2 grid (SplineNet -> nf_distconv, spline stage only)
and the mean-field map is a DistConvertor_ (fused expit-spline-logit kernel).  Both are O(V) or
O(B); nothing here is on the timed hot path.
"""
import copy
import math

import torch

from .._core import Module_
from .modules import SplineNet
from .modules_ import DistConvertor_


def lattice_k2(lat_shape, dtype=None, device=None):
    """hat k^2 = sum_mu 4 sin^2(pi n_mu / N_mu) on the rfftn grid (last axis cut to N/2 + 1),
    fftflow_.py:316-349."""
    total = None
    for mu, n in enumerate(lat_shape):
        k = torch.arange(n, dtype=dtype, device=device) * (2 * math.pi / n)
        k2 = 4 * torch.sin(k / 2) ** 2
        view = [1] * len(lat_shape)
        view[mu] = n
        total = k2.reshape(view) if total is None else total + k2.reshape(view)
    return total[..., :lat_shape[-1] // 2 + 1].contiguous()


def _rescaled_logy(logy, a, ndim):
    """Lattice-spacing rescaling of (log m^2, log kappa) -- fftflow_.py:262-268."""
    la = math.log(a)
    return torch.stack((logy[0] + la * ndim, logy[1] + la * (ndim - 2)))


class IPSD(SplineNet):
    """Inverse power spectral density sigma(k^2) = e^{logy0} + e^{logy1} * spline(k^2 / k^2_max)
    (fftflow_.py:237-273); parameters: the SplineNet logits, then `logy`."""

    def __init__(self, knots_len, *, logy, ignore_zeromode=False, **kwargs):
        super().__init__(knots_len, **kwargs)
        self.logy = torch.nn.Parameter(torch.as_tensor(logy).detach().clone())
        self.ignore_zeromode = ignore_zeromode

    def forward(self, x):
        y = torch.exp(self.logy)
        sigma = y[0] + y[1] * super().forward(x)
        if self.ignore_zeromode:          # the zero mode gets weight 1: no contribution to log J
            hole = torch.zeros_like(sigma)
            hole[(0,) * sigma.dim()] = 1
            sigma = torch.where(hole.bool(), torch.ones_like(sigma), sigma)
        return sigma

    def transfer(self, scale_factor=1, ndim=1):
        new = copy.deepcopy(self)
        with torch.no_grad():
            new.logy.copy_(_rescaled_logy(self.logy, 1 / scale_factor, ndim))
        return new

    @staticmethod
    @torch.no_grad()
    def apply_scale(logy, *, a, ndim):
        return _rescaled_logy(torch.as_tensor(logy), a, ndim)

    @torch.no_grad()
    def infrared_mass(self, max_lat_k2=None):
        return torch.exp(0.5 * self.logy[0])


class FFTNet_(Module_):
    """y = irfftn(rfftn(x) * w), w = sigma(k^2)^(-1/2); log J = sum over ALL modes of log w, i.e. twice the
    sum over the rfftn half minus its first and last columns (fftflow_.py:98-176)."""

    def __init__(self, lat_shape, ipsd_net, ignore_zeromode=False, label='fftnet_'):
        super().__init__(label=label)
        self.lat_shape = tuple(lat_shape)
        self.lat_ndim = len(self.lat_shape)
        self.ipsd_net = ipsd_net
        self.ignore_zeromode = ignore_zeromode
        self.rfft_dim = list(range(-self.lat_ndim, 0))
        k2 = lattice_k2(self.lat_shape)
        self.register_buffer('norm_lat_k2', k2 / k2.max())
        self.register_buffer('max_lat_k2', k2.max())

    @property
    def ipsd(self):
        return self.ipsd_net(self.norm_lat_k2)

    def _weights(self):
        return torch.rsqrt(self.ipsd)

    def _filter(self, x, w):
        spec = torch.fft.rfftn(x, dim=self.rfft_dim)
        return torch.fft.irfftn(spec * w, s=self.lat_shape, dim=self.rfft_dim)

    def forward(self, x, log0=0):
        w = self._weights()
        return self._filter(x, w), log0 + self.log_jacobian(w)

    def backward(self, x, log0=0):
        w = self._weights()
        return self._filter(x, 1 / w), log0 - self.log_jacobian(w)

    def log_jacobian(self, weights):
        lw = torch.log(weights)
        edge = lw[..., 0].sum() + lw[..., -1].sum()     # columns k_last = 0 and the Nyquist (or last) one count once
        return self.create_density(2 * lw.sum() - edge)

    def create_density(self, logj):
        if Module_.propagate_density:
            return (logj / math.prod(self.lat_shape)).expand(self.lat_shape)
        return logj

    @property
    def infrared_mass(self):
        return self.ipsd_net.infrared_mass(self.max_lat_k2)

    def transfer(self, scale_factor=1, shape=None, **extra):
        shape = self.lat_shape if shape is None else shape
        return self.__class__(shape, ipsd_net=self.ipsd_net.transfer(scale_factor=scale_factor, ndim=self.lat_ndim),
                              ignore_zeromode=self.ignore_zeromode)

    @staticmethod
    def build(lat_shape, knots_len=10, eff_mass2=1, eff_kappa=1, a=1, ignore_zeromode=False, nozeromode=False,
              **ipsd_kwargs):
        if nozeromode and not ignore_zeromode:
            raise NotImplementedError("the obsolete `nozeromode` variant is not provided; use ignore_zeromode=True")
        if knots_len < 2:                 # a 2-knot smooth spline is the identity: free-theory spectrum
            knots_len = 2
            ipsd_kwargs.update(dict(smooth=True))
        k2max = float(lattice_k2(lat_shape).max())
        logy = _rescaled_logy(torch.tensor([math.log(eff_mass2), math.log(eff_kappa * k2max)]), a, len(lat_shape))
        return FFTNet_(lat_shape, IPSD(knots_len, logy=logy, ignore_zeromode=ignore_zeromode, **ipsd_kwargs),
                       ignore_zeromode=ignore_zeromode)


class MeanFieldNet_(Module_):
    """Distribution convertor acting on sqrt(V) * (mean of the field) (meanfield_.py:17-66).  With
    `rvol` given, x already IS the mean (one number per sample, broadcastable to the field)."""

    def __init__(self, dc_, label='mean-field'):
        super().__init__(label=label)
        self.dc_ = dc_

    def _run(self, method, x, log0, rvol):
        if rvol is not None:
            new_scaled, log0 = method(x * rvol, log0)
            return new_scaled / rvol, log0
        dims = list(range(1, x.dim()))
        rvol = math.prod(x.shape[1:]) ** 0.5
        mean = x.mean(dim=dims, keepdim=True)
        new_scaled, log0 = method(mean * rvol, log0)
        return x + (new_scaled / rvol - mean), log0

    def forward(self, x, log0=0, rvol=None):
        return self._run(self.dc_.forward, x, log0, rvol)

    def backward(self, x, log0=0, rvol=None):
        return self._run(self.dc_.backward, x, log0, rvol)

    def _hack(self, x, log0=0):
        dims = list(range(1, x.dim()))
        rvol = math.prod(x.shape[1:]) ** 0.5
        mean = x.mean(dim=dims, keepdim=True)
        stack = [(mean.ravel(), log0)]
        scaled, log0 = self.dc_.forward(mean * rvol, log0)
        stack.append((scaled.ravel() / rvol, log0))
        return stack

    @staticmethod
    def build(knots_len=10, **kwargs):
        return MeanFieldNet_(DistConvertor_(knots_len, **kwargs))


class PSDBlock_(Module_):
    """Mean field and fluctuations transformed separately and added back (psd_.py:17-57)."""

    def __init__(self, *, mfnet_, fftnet_, label='psd-block'):
        super().__init__(label=label)
        self.mfnet_ = mfnet_
        self.fftnet_ = fftnet_

    def _parts(self, x, inverse):
        dims = list(range(1, x.dim()))
        rvol = math.prod(x.shape[1:]) ** 0.5
        mean = x.mean(dim=dims, keepdim=True)
        mf = self.mfnet_.backward if inverse else self.mfnet_.forward
        ff = self.fftnet_.backward if inverse else self.fftnet_.forward
        return mean, mf(mean, rvol=rvol), ff(x - mean)

    def forward(self, x, log0=0):
        _, (y_mf, lj_mf), (y_fft, lj_fft) = self._parts(x, False)
        return y_mf + y_fft, log0 + lj_mf + lj_fft

    def backward(self, x, log0=0):
        _, (y_mf, lj_mf), (y_fft, lj_fft) = self._parts(x, True)
        return y_mf + y_fft, log0 + lj_mf + lj_fft

    def _hack(self, x, log0=0):
        mean, (y_mf, lj_mf), (y_fft, lj_fft) = self._parts(x, False)
        return [(mean, log0), (y_mf, lj_mf), (y_fft, lj_fft), (y_mf + y_fft, log0 + lj_mf + lj_fft)]

    def transfer(self, **kwargs):
        return self.__class__(mfnet_=self.mfnet_.transfer(**kwargs), fftnet_=self.fftnet_.transfer(**kwargs))
