"""The flow-network protocol (API of the reference's src/nn/_core.py).

A module whose name ends in an underscore maps `(x, log0) -> (y, log0 + log|det dy/dx|)` in
`forward` and applies the inverse map in `backward`; `log0` may be the python number 0.  Only the
two classes the scalar phi^4 path uses are provided: `Module_` (a leaf transformation) and
`ModuleList_` (their composition).  The reference's multi-channel / invisibility wrappers serve its
gauge-theory scaffolding and are out of scope.
"""
import base64
import copy
import io
import math

import torch


def count_parameters(module):
    return sum(math.prod(p.shape) for p in torch.nn.Module.parameters(module))


class Module_(torch.nn.Module):
    """Leaf of a flow: subclasses implement forward / backward with the Jacobian bookkeeping."""

    propagate_density = False     # True: keep per-site log-densities instead of per-sample sums

    def __init__(self, label=None):
        super().__init__()
        self.label = label

    def forward(self, x, log0=0):
        raise NotImplementedError

    def backward(self, x, log0=0):
        raise NotImplementedError

    def sum_density(self, t):
        """Reduce a per-site log-density to one number per sample (axis 0 is the batch)."""
        if self.propagate_density or t.dim() < 2:
            return t
        return t.flatten(1).sum(dim=1)

    def transfer(self, **kwargs):
        return copy.deepcopy(self)

    npar = property(count_parameters)


def _same_partition(a, b):
    """Do two coupling blocks split the lattice the same way?  (The same mask object, or equal masks: compared once per pair.)"""
    ma, mb = getattr(a, 'mask', None), getattr(b, 'mask', None)
    if ma is None or mb is None:
        return False
    if ma is mb:
        return True
    fn = getattr(ma, 'same_partition', None)
    return bool(fn(mb)) if fn is not None else False


def _run_chain(blocks, method, x, log0):
    """Apply the blocks in order.  Consecutive coupling blocks over the SAME partition hand each other the two parts of the
    field as they are: `cat` followed by the next block's `split` gives the parts back (their supports are disjoint), and on a
    lattice field each of the three is a full pass through memory -- with one coupling block per layer (a common way to
    write a flow; the reference's protocol is unchanged: couplings_.py:54-78) they were ~8 % of a 32^4 inference pass."""
    blocks = list(blocks)
    part_method = 'parts_' + method
    i, n = 0, len(blocks)
    while i < n:
        blk = blocks[i]
        run = getattr(blk, part_method, None)
        j = i + 1
        if run is not None:
            while j < n and getattr(blocks[j], part_method, None) is not None and _same_partition(blk, blocks[j]):
                j += 1
        if run is None or j == i + 1:
            x, log0 = getattr(blk, method)(x, log0)
            i += 1
            continue
        parts = list(blk.mask.split(x))
        for k in range(i, j):
            parts, log0 = getattr(blocks[k], part_method)(parts, log0)
        x = blk.mask.cat(*parts)
        i = j
    return x, log0


class ModuleList_(torch.nn.ModuleList):
    """Composition of flow modules: forward applies them in order, backward inverts them in
    reverse order."""

    _groups = None

    def __init__(self, nets_, label=None):
        super().__init__(nets_)
        self.label = label

    def forward(self, x, log0=0):
        return _run_chain(self, 'forward', x, log0)

    def backward(self, x, log0=0):
        return _run_chain(reversed(list(self)), 'backward', x, log0)

    __call__ = forward        # bypass nn.Module hooks, as the reference does

    def hack(self, x, log0=0):
        """All intermediate (x, log0) pairs of a forward pass, input included."""
        states = [(x, log0)]
        for blk in self:
            states.append(blk.forward(*states[-1]))
        return states

    # -- optimizer parameter groups: [{'ind': [block indices], 'hyper': {...}}, ...]
    def setup_groups(self, groups=None):
        self._groups = groups

    def grouped_parameters(self):
        if self._groups is None:
            return super().parameters()
        return [dict(params=[p for k in grp['ind'] for p in self[k].parameters()], **grp['hyper'])
                for grp in self._groups]

    # -- (de)serialisation helpers
    def get_weights_blob(self):
        buf = io.BytesIO()
        torch.save(self.state_dict(), buf)
        return base64.b64encode(buf.getvalue()).decode('utf-8')

    def set_weights_blob(self, blob, map_location=torch.device('cpu')):
        state = torch.load(io.BytesIO(base64.b64decode(blob.strip())), map_location=map_location,
                           weights_only=True)
        self.load_state_dict(state)

    def _set_trainable(self, flag):
        for p in self.parameters():
            p.requires_grad = flag

    def freeze_parameters(self):
        self._set_trainable(False)

    def unfreeze_parameters(self):
        self._set_trainable(True)

    def transfer(self, **kwargs):
        return type(self)([blk.transfer(**kwargs) for blk in self])

    def to(self, *args, **kwargs):
        for blk in self:
            blk.to(*args, **kwargs)
        return self

    npar = property(count_parameters)
