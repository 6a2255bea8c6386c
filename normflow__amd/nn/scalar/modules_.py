"""Site-local Jacobian-tracking modules (reference: src/nn/scalar/modules_.py).

Expit_, SplineNet_, Logit_ and their composition DistConvertor_ run on the fused K4
kernel (`nf_distconv`): one pass over the field for any subset of the three stages,
the shared spline's knots staged in LDS.  DistConvertor_ keeps the reference's list
structure (so `1.weights_x`-style state_dict keys and `.spline_layer_` still work) but
executes the whole Expit_ -> SplineNet_ -> Logit_ triple as ONE launch.
"""
import math

import torch

from .modules import SplineNet, softplus_ln2
from .._core import Module_, ModuleList_
from ... import _hip


def _as_rows(x):
    """(B, V) view of a field whose axis 0 is the batch."""
    return x.reshape(x.shape[0], -1) if x.dim() > 1 else x.reshape(-1, 1)


def _run_stages(module, x, log0, stages, inverse, knots=None):
    if module is not None and module.propagate_density:
        raise NotImplementedError("propagate_density=True is not provided by the fused kernels")
    v = _as_rows(x)
    l0 = _hip._log0_tensor(log0, v, v.shape[0])
    val, lj = _hip.DistConvFn.apply(v, knots, l0, stages, inverse)
    return val.reshape(x.shape), lj


class Identity_(Module_):
    def __init__(self, label='identity_'):
        super().__init__(label=label)

    def forward(self, x, log0=0, **extra):
        return x, log0

    def backward(self, x, log0=0, **extra):
        return x, log0


class Clone_(Module_):
    def __init__(self, label='clone_'):
        super().__init__(label=label)

    def forward(self, x, log0=0, **extra):
        return x.clone(), log0

    def backward(self, x, log0=0, **extra):
        return x.clone(), log0


class ScaleNet_(Module_):
    """x -> x * softplus_ln2(w), w a single learned logit; log|J| = V log(weight)
    (modules_.py:44-69)."""

    def __init__(self, label='scale_'):
        super().__init__(label=label)
        self._weight = torch.nn.Parameter(torch.zeros(1))

    @property
    def weight(self):
        return softplus_ln2(self._weight)

    def _logj(self, x):
        per_site = torch.log(self.weight)
        if self.propagate_density:
            return per_site.expand(x.shape)
        return (per_site * math.prod(x.shape[1:])).expand(x.shape[0])

    def forward(self, x, log0=0):
        return x * self.weight, log0 + self._logj(x)

    def backward(self, x, log0=0):
        return x / self.weight, log0 - self._logj(x)


class Expit_(Module_):
    """y = 1/(1+e^-x); log|J| = sum(-x + 2 log y) (modules_.py:93-102)."""

    def forward(self, x, log0=0):
        return _run_stages(self, x, log0, _hip.STAGE_EXPIT, False)

    def backward(self, x, log0=0):
        return _run_stages(self, x, log0, _hip.STAGE_LOGIT, False)


class Logit_(Module_):
    """y = log(x/(1-x)); log|J| = -sum log(x(1-x)) (modules_.py:105-114)."""

    def forward(self, x, log0=0):
        return _run_stages(self, x, log0, _hip.STAGE_LOGIT, False)

    def backward(self, x, log0=0):
        return _run_stages(self, x, log0, _hip.STAGE_EXPIT, False)


class SplineNet_(SplineNet, Module_):
    """SplineNet with the log-Jacobian (modules_.py:277-302)."""

    def forward(self, x, log0=0):
        return _run_stages(self, x, log0, _hip.STAGE_SPLINE, False, self.knots())

    def backward(self, x, log0=0):
        return _run_stages(self, x, log0, _hip.STAGE_SPLINE, True, self.knots())


class UnityDistConvertor_(SplineNet_):
    """PDF convertor on [0, 1] (modules_.py:305-316)."""

    def __init__(self, knots_len, symmetric=False, **kwargs):
        extra = dict(xlim=(0.5, 1), ylim=(0.5, 1), extrap={'left': 'anti'}) if symmetric else {}
        super().__init__(knots_len, **kwargs, **extra)


class PhaseDistConvertor_(SplineNet_):
    """PDF convertor on [-pi, pi] (modules_.py:319-330)."""

    def __init__(self, knots_len, symmetric=False, label='phase-dc_', **kwargs):
        pi = math.pi
        extra = (dict(xlim=(0, pi), ylim=(0, pi), extrap={'left': 'anti'}) if symmetric
                 else dict(xlim=(-pi, pi), ylim=(-pi, pi)))
        super().__init__(knots_len, label=label, **kwargs, **extra)


class SgnBiasNet_(Module_):
    """x + sgn(x) w^2: only valid as the very first layer (modules_.py:386-400)."""

    def __init__(self, size=[1], label='sgnbias_'):
        super().__init__(label=label)
        self.w = torch.nn.Parameter(torch.rand(*size) / 10)

    def forward(self, x, log0=0):
        return x + torch.sgn(x) * self.w ** 2, log0

    def backward(self, x, log0=0):
        return x - torch.sgn(x) * self.w ** 2, log0


class DistConvertor_(ModuleList_):
    """PDF convertor for real variables: [SgnBias] [Scale] Expit_ SplineNet_ Logit_ [Scale]
    (modules_.py:333-383); symmetric => spline on (0.5, 1) with an anti-periodic left
    boundary.  The Expit_/SplineNet_/Logit_ triple runs as one fused kernel launch."""

    def __init__(self, knots_len, symmetric=False, label='dc_', sgnbias=False, initial_scale=False,
                 final_scale=False, **kwargs):
        lims = (dict(xlim=(0.5, 1), ylim=(0.5, 1), extrap={'left': 'anti'}) if symmetric
                else dict(xlim=(0, 1), ylim=(0, 1)))
        nets_ = []
        if knots_len > 1:
            nets_ = [Expit_(label='expit_'), SplineNet_(knots_len, label='spline_', **kwargs, **lims),
                     Logit_(label='logit_')]
        if initial_scale:
            nets_.insert(0, ScaleNet_(label='scale_'))
        elif final_scale:
            nets_.append(ScaleNet_(label='scale_'))
        if sgnbias:
            nets_.insert(0, SgnBiasNet_())
        super().__init__(nets_)
        self.label = label

    def _by_label(self, label):
        for net_ in self:
            if net_.label == label:
                return net_

    spline_layer_ = property(lambda self: self._by_label('spline_'))
    scale_layer_ = property(lambda self: self._by_label('scale_'))
    sgnbias_layer_ = property(lambda self: self._by_label('sgnbias_'))

    def _steps(self):
        """Group the children into ('fused', spline_) triples and ('single', module) steps."""
        mods, steps, i = list(self), [], 0
        while i < len(mods):
            tri = mods[i:i + 3]
            if (len(tri) == 3 and type(tri[0]) is Expit_ and isinstance(tri[1], SplineNet_)
                    and type(tri[2]) is Logit_ and not any(t.propagate_density for t in tri)):
                steps.append(('fused', tri[1]))
                i += 3
            else:
                steps.append(('single', mods[i]))
                i += 1
        return steps

    def _run(self, x, log0, inverse):
        steps = self._steps()
        for kind, mod in (reversed(steps) if inverse else steps):
            if kind == 'fused':
                x, log0 = _run_stages(None, x, log0, 7, inverse, mod.knots())
            else:
                x, log0 = mod.backward(x, log0) if inverse else mod.forward(x, log0)
        return x, log0

    def forward(self, x, log0=0):
        return self._run(x, log0, False)

    def backward(self, x, log0=0):
        return self._run(x, log0, True)
