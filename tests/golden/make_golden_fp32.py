#!/usr/bin/env python3
"""Reference-fp32 noise floor: re-run THE REFERENCE in float32 on the inputs of the committed fp64 goldens.

The fp64 fixtures (make_golden.py) are the truth a kernel is compared with.  north_star's bound on an fp32 kernel is
"1e-5 relative"; where the reference's OWN fp32 arithmetic is further than that from its fp64 result (ill-conditioned
splines, wide logits, the conv in front of a block), a GPU test may loosen its bound to `max(1e-5, 2 * err_ref_fp32)` --
and only there.  This script measures err_ref_fp32 instead of assuming it: for every case of atoms / distconv / blocks /
callers it loads the fp64 inputs from the committed .npz, casts inputs and weights to float32, runs the reference with
torch's default dtype set to float32 (what a user running the reference in single precision does; it also keeps the
reference's default-dtype `zeros`/`empty` calls, SURVEY App. A #3-4, from silently promoting to fp64), and stores the
reference's float32 OUTPUTS (y, logJ, gradients, inverse) as float32 arrays in tests/golden/ref_fp32.npz.

Container-only (needs /root/reference), same recipe as make_golden.py:
    mkdir -p /tmp/nf_oracle && ln -sfn /root/reference/src /tmp/nf_oracle/normflow
    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/tmp/nf_oracle:tests python3 tests/golden/make_golden_fp32.py
The fixture is data only (the reference's numerical outputs), never reference source.
"""
import os
import sys
import warnings

import numpy as np

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import torch  # noqa: E402
import normflow  # noqa: E402,F401  (the REFERENCE)
from normflow.mask import EvenOddMask  # noqa: E402
from normflow.nn import (AffineCoupling_, ShiftCoupling_, RQSplineCoupling_,  # noqa: E402
                         MultiRQSplineCoupling_, DistConvertor_, ConvAct)
from normflow.action import ScalarPhi4Action  # noqa: E402
from normflow.prior import NormalPrior  # noqa: E402

torch.set_default_device('cpu')
torch.set_default_dtype(torch.float32)
F = torch.float32

ATOM_OPTS = {
    "rqs_lin": dict(xlim=(-2.0, 2.0), ylim=(-2.5, 1.5), extrap={'left': 'linear', 'right': 'linear'}),
    "rqs_anti": dict(xlim=(0.0, 2.0), ylim=(0.0, 2.0), extrap={'left': 'anti', 'right': 'linear'}),
    "rqs_none": dict(xlim=(0.0, 1.0), ylim=(0.0, 1.0), extrap={}),
    "rqs_onesided": dict(xlim=(0.0, 1.0), ylim=(0.0, 1.0), extrap={'right': 'linear'}),
    "rqs_fixedx": dict(xlim=(-1.0, 1.0), ylim=(-1.0, 1.0), extrap={'left': 'linear', 'right': 'linear'}),
    "multirqs": dict(xlims=[(-2.0, 2.0), (-1.0, 3.0)], ylims=[(-2.0, 2.0), (-3.0, 1.0)],
                     extraps=[{'left': 'linear', 'right': 'linear'}] * 2),
}


class Fixed(torch.nn.Module):
    def __init__(self, out):
        super().__init__()
        self.out = out

    def forward(self, x):
        return self.out


def T(a):
    return torch.from_numpy(np.asarray(a)).to(F)


def npy(t):
    return t.detach().cpu().numpy().astype(np.float32)


def gen_atoms(store):
    z = np.load(os.path.join(HERE, "atoms.npz"))
    for tag in [str(c) for c in z["_cases"]]:
        kind = tag.split("/")[0]
        g = lambda k: T(z[f"{tag}/{k}"])
        shape, parity = tuple(int(v) for v in z[f"{tag}/shape"]), int(z[f"{tag}/parity"])
        mask = EvenOddMask(shape=shape)
        out = g("out").requires_grad_(True)
        net = Fixed(out)
        if kind == "affine":
            cpl = AffineCoupling_([net], mask=mask)
        elif kind == "shift":
            cpl = ShiftCoupling_([net], mask=mask)
        elif kind == "multirqs":
            cpl = MultiRQSplineCoupling_([net], mask=mask, **ATOM_OPTS[kind])
        else:
            kw = dict(ATOM_OPTS[kind])
            if kind == "rqs_fixedx":
                kw["knots_x"] = g("knots_x")
            cpl = RQSplineCoupling_([net], mask=mask, **kw)
        x_active = g("x_active").requires_grad_(True)
        x_frozen = torch.zeros_like(x_active)        # Fixed ignores it
        y, logJ = cpl.atomic_forward(x_active=x_active, x_frozen=x_frozen, parity=parity, net=net, log0=g("log0"))
        loss = logJ.mean() + (y ** 2).mean()
        gx, gout = torch.autograd.grad(loss, (x_active, out), allow_unused=True)
        with torch.no_grad():
            # inverse from the fp64 golden's y (the GPU tests invert the golden y as well)
            xhat, lrt = cpl.atomic_backward(x_active=g("y"), x_frozen=x_frozen, parity=parity, net=net, log0=g("logJ"))
        assert y.dtype == F and logJ.dtype == F, (tag, y.dtype, logJ.dtype)
        store[f"atoms/{tag}/y"] = npy(y)
        store[f"atoms/{tag}/logJ"] = npy(logJ)
        store[f"atoms/{tag}/grad_x"] = npy(gx if gx is not None else torch.zeros_like(x_active))
        store[f"atoms/{tag}/grad_out"] = npy(gout if gout is not None else torch.zeros_like(out))
        store[f"atoms/{tag}/xhat"] = npy(xhat)
        store[f"atoms/{tag}/logJ_rt"] = npy(lrt)


def gen_distconv(store):
    z = np.load(os.path.join(HERE, "distconv.npz"))
    for tag in [str(c) for c in z["_cases"]]:
        g = lambda k: T(z[f"{tag}/{k}"])
        sym, smooth = "sym1" in tag, "sm1" in tag
        m = int(tag.split("m")[-1])
        net_ = DistConvertor_(m, symmetric=sym, smooth=smooth)
        sp = net_.spline_layer_
        with torch.no_grad():
            sp.weights_x.copy_(g("wx"))
            sp.weights_y.copy_(g("wy"))
            if not smooth:
                sp.weights_d.copy_(g("wd"))
        net_.to(F)
        x = g("x").requires_grad_(True)
        y, logJ = net_(x, g("log0"))
        loss = logJ.mean() + (y ** 2).mean()
        params = [sp.weights_x, sp.weights_y] + ([] if smooth else [sp.weights_d])
        grads = torch.autograd.grad(loss, [x] + params)
        with torch.no_grad():
            xhat, lrt = net_.backward(g("y"), g("logJ"))
        assert y.dtype == F
        store[f"distconv/{tag}/y"] = npy(y)
        store[f"distconv/{tag}/logJ"] = npy(logJ)
        store[f"distconv/{tag}/xhat"] = npy(xhat)
        store[f"distconv/{tag}/logJ_rt"] = npy(lrt)
        for name, gr in zip(["grad_x", "grad_wx", "grad_wy", "grad_wd"], grads):
            store[f"distconv/{tag}/{name}"] = npy(gr)


def gen_blocks(store):
    z = np.load(os.path.join(HERE, "blocks.npz"))
    for tag in [str(c) for c in z["_cases"]]:
        kind, dd = tag.split("/")
        d = int(dd[1:])
        shape = tuple(int(v) for v in z[f"{tag}/shape"])
        n_out = 2 if kind == "affine" else 16
        nets = [ConvAct(1, n_out, 3, conv_dim=d, hidden_sizes=[4, 4], acts=['tanh', 'tanh', None]) for _ in range(3)]
        mask = EvenOddMask(shape=shape)
        if kind == "affine":
            cpl = AffineCoupling_(nets, mask=mask)
        else:
            cpl = RQSplineCoupling_(nets, mask=mask, xlim=(-3.0, 3.0), ylim=(-3.0, 3.0),
                                    extrap={'left': 'linear', 'right': 'linear'})
        names = [n for n, _ in cpl.named_parameters()]
        with torch.no_grad():
            for n, p in cpl.named_parameters():
                p.copy_(T(z[f"{tag}/param/{n}"]))
        cpl.to(F)
        x = T(z[f"{tag}/x"]).requires_grad_(True)
        y, logJ = cpl(x)
        loss = logJ.mean() + (y ** 2).mean()
        plist = list(cpl.parameters())
        grads = torch.autograd.grad(loss, [x] + plist)
        assert y.dtype == F
        store[f"blocks/{tag}/y"] = npy(y)
        store[f"blocks/{tag}/logJ"] = npy(logJ)
        store[f"blocks/{tag}/grad_x"] = npy(grads[0])
        for n, gp in zip(names, grads[1:]):
            store[f"blocks/{tag}/gparam/{n}"] = npy(gp)


def gen_callers(store):
    z = np.load(os.path.join(HERE, "callers.npz"))
    net_ = DistConvertor_(knots_len=10, symmetric=True)
    sp = net_.spline_layer_
    with torch.no_grad():
        sp.weights_x.copy_(T(z["c1/wx"]))
        sp.weights_y.copy_(T(z["c1/wy"]))
        sp.weights_d.copy_(T(z["c1/wd"]))
    net_.to(F)
    action = ScalarPhi4Action(kappa=0, m_sq=-1.2, lambd=0.5)
    with torch.no_grad():
        y, logJ = net_(T(z["c1/x"]))
        store["callers/c1/y"] = npy(y)
        store["callers/c1/logJ"] = npy(logJ)
        store["callers/c1/logp"] = npy(-action(y))
    kap, msq, lam = (float(v) for v in z["phi4/coef"])
    for d in (1, 2, 3, 4):
        cfg = T(z[f"phi4/d{d}/cfg"])
        act = ScalarPhi4Action(kappa=kap, m_sq=msq, lambd=lam)
        store[f"callers/phi4/d{d}/S"] = npy(act(cfg))
        store[f"callers/phi4/d{d}/logr"] = npy(NormalPrior(shape=tuple(cfg.shape[1:])).log_prob(cfg))


if __name__ == "__main__":
    store = {}
    gen_atoms(store)
    gen_distconv(store)
    gen_blocks(store)
    gen_callers(store)
    path = os.path.join(HERE, "ref_fp32.npz")
    np.savez_compressed(path, **store)
    print(f"ref_fp32: {os.path.getsize(path)/1024:.1f} KiB, keys={len(store)}")
