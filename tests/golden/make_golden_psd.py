#!/usr/bin/env python3
"""Golden vectors for the spectral block (PSDBlock_ = MeanFieldNet_ + FFTNet_), by RUNNING THE REFERENCE.

Container-only, like make_golden.py:

    mkdir -p /tmp/nf_oracle && ln -sfn /root/reference/src /tmp/nf_oracle/normflow
    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/tmp/nf_oracle python3 tests/golden/make_golden_psd.py

The reference calls `np.product`, which NumPy 2 removed; this script restores the alias
(`np.product = np.prod`, identical function in NumPy 1.x) in ITS OWN process before importing the reference.
Nothing of the reference is modified or copied; the fixture holds inputs, parameters and outputs only.
"""
import os
import warnings

import numpy as np

if not hasattr(np, "product"):
    np.product = np.prod          # NumPy-1 alias the reference relies on

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))

import torch  # noqa: E402
import normflow  # noqa: E402,F401  (the REFERENCE)
from normflow.nn import FFTNet_, MeanFieldNet_, PSDBlock_  # noqa: E402

torch.set_default_device('cpu')
assert torch.get_default_dtype() == torch.float64

out = {}


def perturb(module, seed, scale=0.4):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            p.add_(scale * torch.randn(p.shape, generator=g))


def record(tag, net, x):
    for k, v in net.state_dict().items():
        out[f"{tag}/state/{k}"] = v.detach().numpy().copy()
    out[f"{tag}/x"] = x.numpy().copy()
    xr = x.clone().requires_grad_(True)
    l0 = torch.linspace(-0.3, 0.4, x.shape[0])
    y, lj = net.forward(xr, l0)
    loss = lj.mean() + (y ** 2).mean()
    grads = torch.autograd.grad(loss, [xr] + list(net.parameters()))
    out[f"{tag}/log0"] = l0.numpy().copy()
    out[f"{tag}/y"] = y.detach().numpy().copy()
    out[f"{tag}/logJ"] = lj.detach().numpy().copy()
    out[f"{tag}/grad_x"] = grads[0].numpy().copy()
    for (name, _), gp in zip(net.named_parameters(), grads[1:]):
        out[f"{tag}/grad/{name}"] = gp.numpy().copy()
    with torch.no_grad():
        xb, lb = net.backward(y.detach(), lj.detach())
    out[f"{tag}/xb"] = xb.numpy().copy()
    out[f"{tag}/lb"] = lb.numpy().copy()


cases = [
    ("psd2d", (8, 8), dict(knots_len=6, symmetric=True, final_scale=True, smooth=True), dict(knots_len=5, ignore_zeromode=True)),
    ("psd3d", (4, 6, 4), dict(knots_len=4, symmetric=False, smooth=False), dict(knots_len=4, ignore_zeromode=False)),
    ("psd1d_odd", (9,), dict(knots_len=5, symmetric=True, smooth=True), dict(knots_len=1, ignore_zeromode=True, eff_mass2=0.7, eff_kappa=1.3, a=0.5)),
]
for i, (tag, shape, mfdict, fftdict) in enumerate(cases):
    torch.manual_seed(100 + i)
    mf = MeanFieldNet_.build(**mfdict)
    fft = FFTNet_.build(shape, **fftdict)
    blk = PSDBlock_(mfnet_=mf, fftnet_=fft)
    perturb(blk, 7 + i)
    x = 0.8 * torch.randn(5, *shape)
    record(tag, blk, x)
    # the parts on their own
    record(tag + "_fft", fft, x)
    record(tag + "_mf", mf, x)
    out[f"{tag}/k2norm"] = fft.norm_lat_k2.numpy().copy()
    out[f"{tag}/k2max"] = np.array(float(fft.max_lat_k2))
    out[f"{tag}/ipsd"] = fft.ipsd.detach().numpy().copy()
    out[f"{tag}/ir_mass"] = np.array(float(fft.infrared_mass))

np.savez_compressed(os.path.join(HERE, "psd.npz"), **out)
print("wrote psd.npz with", len(out), "arrays")
