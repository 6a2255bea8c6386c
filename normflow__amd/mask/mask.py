"""Site masks for coupling layers (reference: src/mask/mask.py).

A mask owns two uint8 buffers, `_mask` and `_c_mask` (= 1 - `_mask`), registered under
the reference's names so that state_dict()s interchange.  Besides the reference's
`split / cat / purify` (kept as plain tensor ops for API compatibility), a mask hands
the HIP kernels the *activity bytes* of a channel and tells them whether the aligned
site pairs (2h, 2h+1) each hold exactly one active site -- the condition for the
HBM-efficient pair layout of the per-site parameters (include/normflow_hip.h).
"""
import torch


def _index_sum(shape, skip=None):
    """sum_mu ind[mu] over a lattice of `shape`, as an int64 tensor (vectorised: the
    reference's O(V) python loop, mask.py:55-60, takes 10 s at 32^4)."""
    total = torch.zeros(tuple(shape), dtype=torch.int64, device='cpu')
    for mu, n in enumerate(shape):
        if mu == skip:
            continue
        view = [1] * len(shape)
        view[mu] = n
        total = total + torch.arange(n, dtype=torch.int64, device='cpu').reshape(view)
    return total


class Mask(torch.nn.Module):
    """0/1 partition of the lattice sites into channel 0 (`_mask`) and channel 1."""

    def __init__(self, **mask_kwargs):
        super().__init__()
        # like the reference (mask.py:17-28) the buffers live on torch's DEFAULT device -- 'cuda' once the package is
        # imported on a GPU box -- so that a net assembled from defaults meets the prior's samples on the same device
        m = self.make_mask(**mask_kwargs).to(device=torch.get_default_device(), dtype=torch.uint8)
        self.register_buffer('_mask', m)
        self.register_buffer('_c_mask', 1 - m)
        self.mask_kwargs = mask_kwargs
        self._pairable = None

    def __str__(self):
        return str(self._mask)

    # -- reference protocol (mask.py:30-37)
    def split(self, x):
        return self._mask * x, self._c_mask * x

    def cat(self, x_0, x_1):
        return x_0 + x_1

    def purify(self, x_chnl, channel):
        return x_chnl * (self._mask if channel == 0 else self._c_mask)

    def same_partition(self, other):
        """Does `other` split the lattice into the same two channels?  (Compared once per pair of mask objects.)"""
        if other is self:
            return True
        m = getattr(other, '_mask', None)
        if m is None or type(other) is not type(self) or m.shape != self._mask.shape:
            return False
        seen = self.__dict__.setdefault('_same_as', {})
        key = (id(other), m.data_ptr(), self._mask.data_ptr())
        if key not in seen:
            seen[key] = bool(torch.equal(m.to(self._mask.device), self._mask))
        return seen[key]

    # -- kernel-side view
    def activity(self, channel):
        """uint8 tensor over the lattice: 1 where `channel` is the active partition."""
        return self._mask if channel == 0 else self._c_mask

    @property
    def pairable(self):
        """True if every aligned pair of consecutive sites holds one site per channel."""
        if self._pairable is None:
            flat = self._mask.reshape(-1)
            ok = flat.numel() % 2 == 0 and flat.numel() > 0
            if ok:
                ok = bool((flat.reshape(-1, 2).sum(dim=1) == 1).all().item())
            self._pairable = ok
        return self._pairable

    def checkerboard_parity(self, channel):
        """a in {0, 1} if `channel` is exactly the set of sites with coordinate sum = a (mod 2)
        (a plain even-odd mask), else None.  Lets the parameter net emit its output at the
        active sites only."""
        if not hasattr(self, '_cb'):
            m = self._mask.cpu()
            par = (_index_sum(m.shape) % 2).to(torch.uint8)
            self._cb = 0 if torch.equal(m, 1 - par) else (1 if torch.equal(m, par) else None)
        if self._cb is None:
            return None
        return self._cb if channel == 0 else 1 - self._cb

    @staticmethod
    def make_mask(**kwargs):
        raise NotImplementedError


class EvenOddMask(Mask):
    """Checkerboard: mask = (1 - parity + sum(ind)) % 2; with `exclude_mu` the parity
    ignores that axis (mask.py:46-61)."""

    @staticmethod
    def make_mask(*, shape, parity=0, exclude_mu=None):
        return ((1 - parity + _index_sum(shape, skip=exclude_mu)) % 2).to(torch.uint8)


class AlongAxesEvenOddMask(Mask):
    """Alternates along one axis only (mask.py:64-72)."""

    @staticmethod
    def make_mask(*, shape, parity=0, mu=0):
        total = _index_sum(shape)
        only = total - _index_sum(shape, skip=mu)
        return ((1 - parity + only) % 2).to(torch.uint8)


class DummyMask:
    """All sites in one channel (mask.py:75-94)."""

    def __init__(self, parity=0):
        self.parity = parity

    def split(self, x):
        return (x, None) if self.parity == 0 else (None, x)

    def cat(self, x_0, x_1):
        return x_0 if self.parity == 0 else x_1

    @staticmethod
    def purify(x_chnl, *args, **kwargs):
        return x_chnl
