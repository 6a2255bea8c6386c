"""Flow-network protocol (reference: src/nn/_core.py).

A trailing underscore marks modules whose forward/backward thread the log-Jacobian:
    forward(x, log0=0) -> (y, log0 + log|det dy/dx|),  backward = the inverse map.
"""
import base64
import copy
import io
import math

import torch


def _count(params):
    return sum(math.prod(p.shape) for p in params)


class Module_(torch.nn.Module):
    """Base of all Jacobian-tracking modules (nn/_core.py:12-42)."""

    propagate_density = False

    def __init__(self, label=None):
        super().__init__()
        self.label = label

    def forward(self, x, log0=0):
        pass

    def backward(self, x, log0=0):
        pass

    def transfer(self, **kwargs):
        return copy.deepcopy(self)

    @property
    def npar(self):
        return _count(self.parameters())

    def sum_density(self, x):
        """Per-sample sum over all non-batch axes unless densities are propagated."""
        if self.propagate_density:
            return x
        return x.sum(dim=tuple(range(1, x.dim()))) if x.dim() > 1 else x


class ModuleList_(torch.nn.ModuleList):
    """Sequential composition of Module_s (nn/_core.py:46-134)."""

    _groups = None

    def __init__(self, nets_, label=None):
        super().__init__(nets_)
        self.label = label

    def forward(self, x, log0=0):
        for net_ in self:
            x, log0 = net_.forward(x, log0)
        return x, log0

    def backward(self, x, log0=0):
        for net_ in reversed(list(self)):
            x, log0 = net_.backward(x, log0)
        return x, log0

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    def hack(self, x, log0=0):
        """forward() that also returns every intermediate (x, log0)."""
        trace = [(x, log0)]
        for net_ in self:
            x, log0 = net_.forward(x, log0)
            trace.append((x, log0))
        return trace

    def setup_groups(self, groups=None):
        """groups: [{'ind': [block indices], 'hyper': {optimizer kwargs}}, ...]"""
        self._groups = groups

    def grouped_parameters(self):
        if self._groups is None:
            return super().parameters()
        out = []
        for grp in self._groups:
            params = [p for k in grp['ind'] for p in self[k].parameters()]
            out.append(dict(params=params, **grp['hyper']))
        return out

    def transfer(self, **kwargs):
        return self.__class__([net_.transfer(**kwargs) for net_ in self])

    def get_weights_blob(self):
        buf = io.BytesIO()
        torch.save(self.state_dict(), buf)
        return base64.b64encode(buf.getbuffer()).decode('utf-8')

    def set_weights_blob(self, blob, map_location=torch.device('cpu')):
        raw = io.BytesIO(base64.b64decode(blob.strip()))
        self.load_state_dict(torch.load(raw, map_location=map_location, weights_only=True))

    def freeze_parameters(self):
        for p in self.parameters():
            p.requires_grad = False

    def unfreeze_parameters(self):
        for p in self.parameters():
            p.requires_grad = True

    @property
    def npar(self):
        return _count(super().parameters())

    def to(self, *args, **kwargs):
        for net_ in self:
            net_.to(*args, **kwargs)
        return self


class MultiChannelModule_(torch.nn.ModuleList):
    """One Module_ per data channel (nn/_core.py:138-183)."""

    def __init__(self, nets_, label=None, channels_axis=1, keep_channels_axis=True):
        super().__init__(nets_)
        self.channels_axis = channels_axis
        self.keep_channels_axis = keep_channels_axis
        self.label = label

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    def forward(self, x, log0=0):
        return self._map(x, [n.forward for n in self], log0)

    def backward(self, x, log0=0):
        return self._map(x, [n.backward for n in self], log0)

    def _map(self, x, fns, log0=0):
        ax = self.channels_axis
        parts = x.split(1, dim=ax) if self.keep_channels_axis else x.unbind(dim=ax)
        assert len(parts) == len(fns), "mismatch in channels of input & network."
        outs = [f(p) for f, p in zip(fns, parts)]
        join = torch.cat if self.keep_channels_axis else torch.stack
        return join([o[0] for o in outs], dim=ax), log0 + sum(o[1] for o in outs)

    @property
    def npar(self):
        return _count(super().parameters())


class MultiOutChannelModule_(MultiChannelModule_):
    """Every sub-module sees the whole input; outputs are concatenated (nn/_core.py:187-195)."""

    def _map(self, x, fns, log0=0):
        outs = [f(x) for f in fns]
        return torch.cat([o[0] for o in outs], dim=self.channels_axis), log0 + sum(o[1] for o in outs)


class InvisibilityMaskWrapperModule_(Module_):
    """Hide part of the input from `net_` (nn/_core.py:199-231)."""

    def __init__(self, net_, *, mask):
        super().__init__(label=f'wrapper:{net_.label}')
        self.net_ = net_
        self.mask = mask
        self.net_.propagate_density = True

    def _apply_visible(self, fn, x, log0):
        vis, hidden = self.mask.split(x)
        vis, dens = fn(vis)
        vis = self.mask.purify(vis, channel=0)
        logJ = self.sum_density(self.mask.purify(dens, channel=0))
        return self.mask.cat(vis, hidden), log0 + logJ

    def forward(self, x, log0=0):
        return self._apply_visible(self.net_.forward, x, log0)

    def backward(self, x, log0=0):
        return self._apply_visible(self.net_.backward, x, log0)
