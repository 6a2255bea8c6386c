"""N-dimensional circular 'same' convolution (reference: src/nn/scalar/convNd.py).

Parameters are stored exactly as the reference stores them -- `_conv_lower_dim.weight`
of shape (out*k0, in, k1..k_{N-1}) plus a separate `bias` (randn-initialised,
convNd.py:79-84) -- so state_dict()s interchange.  The forward pass is this package's
own decomposition: the leading lattice axis is folded into the batch, the lower-
dimensional lattice is wrap-padded ONCE, and each of the k0 kernel slices is applied as
an unpadded (N-1)-d convolution whose output is rolled by its offset and accumulated.
"""
from typing import Tuple, Union

import torch
import torch.nn.functional as F

_CONV = {1: F.conv1d, 2: F.conv2d, 3: F.conv3d}


def wrap_pad(x, pads):
    """Circularly pad the trailing len(pads) axes of x by pads[i] on both sides."""
    first = x.dim() - len(pads)
    for ax, p in enumerate(pads, start=first):
        if p:
            n = x.shape[ax]
            x = torch.cat((x.narrow(ax, n - p, p), x, x.narrow(ax, 0, p)), dim=ax)
    return x


def circular_conv(x, weight, bias=None, force_torch=False):
    """Circular 'same' cross-correlation for 1..4 lattice dimensions.
    x: (B, Cin, *L); weight: (Cout, Cin, *k), k odd; bias: (Cout,) | None.

    fp32 tensors on the GPU go to the MFMA kernel (nf_conv_fwd); other dtypes / devices,
    and the kernel's own VJP, use the torch-op decomposition below."""
    from ... import _hip
    if not force_torch and _hip.conv_supported(x, weight):
        return _hip.conv_layer(x, weight, bias)
    d = x.dim() - 2
    ks = tuple(weight.shape[2:])
    if d <= 3:
        return _CONV[d](wrap_pad(x, [k // 2 for k in ks]), weight, bias)
    if d != 4:
        raise NotImplementedError("lattice dimensions above 4 are not supported")
    B, Ci, L0 = x.shape[:3]
    rest = tuple(x.shape[3:])
    folded = wrap_pad(x.movedim(2, 1).reshape(B * L0, Ci, *rest), [k // 2 for k in ks[1:]])
    acc = None
    for j in range(ks[0]):
        part = F.conv3d(folded, weight[:, :, j]).reshape(B, L0, weight.shape[0], *rest)
        part = torch.roll(part, ks[0] // 2 - j, dims=1)
        acc = part if acc is None else acc + part
    acc = acc.movedim(1, 2)
    if bias is not None:
        acc = acc + bias.reshape(1, -1, 1, 1, 1, 1)
    return acc.contiguous()


class ConvNd(torch.nn.Module):
    """N-d circular convolution whose parameters live in an (N-1)-d conv module."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: Union[Tuple[int, ...], int], *,
                 conv_ndim: int, stride: int = 1, padding: Union[Tuple[int, ...], int, str] = 'same',
                 padding_mode: str = 'circular', dilation: int = 1, groups: int = 1, bias: bool = True,
                 device=None, dtype=None):
        super().__init__()
        assert conv_ndim > 1, "conv_ndim must be larger than 1."
        assert stride == 1 and dilation == 1 and groups == 1, "only stride=dilation=groups=1"
        assert padding_mode == 'circular', "only circular padding"
        if isinstance(kernel_size, int):
            kernel_size = [kernel_size] * conv_ndim
        if isinstance(padding, int):
            padding = (padding,) * conv_ndim
        if isinstance(padding, tuple):
            assert all(p == k // 2 for p, k in zip(padding, kernel_size)), "only 'same' padding"
        else:
            assert padding == 'same', "only 'same' padding"
        assert 2 <= conv_ndim <= 4, "conv_ndim must be 2, 3 or 4"
        self.conv_ndim = conv_ndim
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = list(kernel_size)
        lower = {4: torch.nn.Conv3d, 3: torch.nn.Conv2d, 2: torch.nn.Conv1d}[conv_ndim]
        # only a parameter container (same shapes and default init as the reference's)
        self._conv_lower_dim = lower(in_channels, out_channels * kernel_size[0], tuple(kernel_size[1:]),
                                     padding='same', padding_mode='circular', bias=False,
                                     device=device, dtype=dtype)
        self.bias = torch.nn.Parameter(torch.randn(out_channels, dtype=dtype, device=device)) if bias else None

    @property
    def weight(self):
        """Standard layout (out, in, k0, ..., k_{N-1})."""
        w = self._conv_lower_dim.weight
        return w.reshape(self.out_channels, self.kernel_size[0], self.in_channels,
                         *self.kernel_size[1:]).movedim(1, 2)

    def forward(self, input):
        if input.dim() == self.conv_ndim + 1:
            input = input.unsqueeze(0)
        assert input.dim() == self.conv_ndim + 2 and input.shape[1] == self.in_channels, \
            "Inconsistant input shape"
        return circular_conv(input, self.weight, self.bias)


class Conv4d(ConvNd):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, conv_ndim=4, **kwargs)
