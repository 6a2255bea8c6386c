#!/usr/bin/env python3
"""Golden vectors for the small-lattice fused kernel (nf_conv_s.hip), produced by RUNNING THE REFERENCE: whole Coupling_
blocks (ConvAct nets + affine / RQ-spline coupling, forward and inverse) on 2-D and 3-D lattices whose fastest axis has 16
sites -- the shapes that kernel takes.  Container-only (needs /root/reference); the fixture it writes is data only.

    mkdir -p /tmp/nf_oracle && ln -sfn /root/reference/src /tmp/nf_oracle/normflow
    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/tmp/nf_oracle python3 tests/golden/make_golden_small16.py
"""
import os
import warnings

import numpy as np

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))

import torch  # noqa: E402
import normflow  # noqa: E402,F401  (the REFERENCE; sets default dtype fp64)
from normflow.mask import EvenOddMask  # noqa: E402
from normflow.nn import AffineCoupling_, RQSplineCoupling_, ConvAct  # noqa: E402

torch.set_default_device('cpu')
assert torch.get_default_dtype() == torch.float64
npy = lambda t: t.detach().cpu().numpy()

CASES = [  # tag, kind, lattice, knots_len, hidden, activations, batch
    ("affine/16x16", "affine", (16, 16), 0, 8, ('tanh', 'tanh'), 4),        # BASELINE config 2's lattice and net
    ("affine/2x4x16", "affine", (2, 4, 16), 0, 4, ('tanh', 'tanh'), 3),
    ("rqs/4x16", "rqs", (4, 16), 8, 8, ('tanh', 'tanh'), 3),
    ("rqs/4x6x16", "rqs", (4, 6, 16), 16, 8, ('tanh', 'tanh'), 3),          # BASELINE config 3's net on a smaller 3-D lattice
]
LIM = dict(xlim=(-3.0, 3.0), ylim=(-3.0, 3.0), extrap={'left': 'linear', 'right': 'linear'})

store, meta = {}, []
for i, (tag, kind, shape, m, hidden, acts, B) in enumerate(CASES):
    torch.manual_seed(7000 + i)
    d = len(shape)
    n_out = 2 if kind == "affine" else 3 * m - 2
    nets = [ConvAct(1, n_out, 3, conv_dim=d, hidden_sizes=[hidden, hidden], acts=[acts[0], acts[1], None]) for _ in range(2)]
    with torch.no_grad():
        for net in nets:
            for p in list(net.parameters())[-2:]:
                p.mul_(0.3)
    mask = EvenOddMask(shape=shape)
    cpl = AffineCoupling_(nets, mask=mask) if kind == "affine" else RQSplineCoupling_(nets, mask=mask, **LIM)
    x = 1.3 * torch.randn((B,) + shape)
    with torch.no_grad():
        y, logJ = cpl(x)
        xhat, logJ_rt = cpl.backward(y, logJ)
    meta.append(tag)
    store.update({f"{tag}/x": npy(x), f"{tag}/y": npy(y), f"{tag}/logJ": npy(logJ), f"{tag}/xhat": npy(xhat),
                  f"{tag}/logJ_rt": npy(logJ_rt), f"{tag}/shape": np.array(shape), f"{tag}/m": np.array(m),
                  f"{tag}/hidden": np.array(hidden)})
    for n, p in cpl.named_parameters():
        store[f"{tag}/param/{n}"] = npy(p)
store['_cases'] = np.array(meta)
path = os.path.join(HERE, "small16.npz")
np.savez_compressed(path, **store)
print(f"small16: {os.path.getsize(path) / 1024:.1f} KiB, {len(meta)} cases")
