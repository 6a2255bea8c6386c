#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Container-only: needs /root/reference (which never travels to the GPU box).  It is
committed so that the fixtures can be regenerated and audited; the fixtures it
writes are data only (inputs and the reference's outputs), never reference source.

    mkdir -p /tmp/nf_oracle && ln -sfn /root/reference/src /tmp/nf_oracle/normflow
    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/tmp/nf_oracle python3 tests/golden/make_golden.py

(the reference's package dir is `src/`, mapped to `normflow` by its setup.py:29-44).
All tensors are fp64 (the reference's default dtype, src/device/__init__.py:13),
CPU.  Cases follow SURVEY.md section 8(c).
"""
import os
import sys
import warnings

import numpy as np

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))

import torch  # noqa: E402
import normflow  # noqa: E402  (the REFERENCE; sets default dtype fp64)
from normflow.mask import EvenOddMask  # noqa: E402
from normflow.nn import (AffineCoupling_, ShiftCoupling_, RQSplineCoupling_,  # noqa: E402
                         MultiRQSplineCoupling_, DistConvertor_, ConvAct, ModuleList_)
from normflow.action import ScalarPhi4Action  # noqa: E402
from normflow.prior import NormalPrior  # noqa: E402
from normflow import Model  # noqa: E402

torch.set_default_device('cpu')
assert torch.get_default_dtype() == torch.float64


class Fixed(torch.nn.Module):
    """A stand-in 'net' that returns a precomputed output tensor."""

    def __init__(self, out):
        super().__init__()
        self.out = out

    def forward(self, x):
        return self.out


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB, keys={len(arrays)}")


# ----------------------------------------------------------------------------- atoms
def atom_case(cpl_cls, shape, B, n_ch, parity, seed, xscale=1.0, oscale=1.0, x_uniform=None,
              x_channels=None, knot_hits=False, **cpl_kw):
    g = torch.Generator().manual_seed(seed)
    mask = EvenOddMask(shape=shape)
    xshape = (B,) + ((x_channels,) if x_channels else ()) + tuple(shape)
    if x_uniform is not None:
        lo, hi = x_uniform
        x = lo + (hi - lo) * torch.rand(xshape, generator=g)
    else:
        x = xscale * torch.randn(xshape, generator=g)
    out = oscale * torch.randn((B, n_ch) + tuple(shape), generator=g)
    net = Fixed(out)
    cpl = cpl_cls([net], mask=mask, **cpl_kw)
    if knot_hits and isinstance(cpl, RQSplineCoupling_):
        # put some inputs exactly on knots (searchsorted tie semantics) and on the limits
        with torch.no_grad():
            sp = cpl.make_spline(out)
            kx = sp.knots_x
            flat = x.reshape(B, -1)
            kflat = kx.reshape(B, kx.shape[1], -1)
            for j in range(0, flat.shape[1], 3):
                flat[:, j] = kflat[:, (j // 3) % kx.shape[1], j]
            x = flat.reshape(x.shape)
    x_active = mask.purify(x, channel=parity).clone().requires_grad_(True)
    x_frozen = mask.purify(x, channel=1 - parity)
    out.requires_grad_(True)
    log0 = 0.1 * torch.randn(B, generator=g)
    y, logJ = cpl.atomic_forward(x_active=x_active, x_frozen=x_frozen, parity=parity, net=net, log0=log0)
    loss = logJ.mean() + (y ** 2).mean()
    gx, gout = torch.autograd.grad(loss, (x_active, out))
    with torch.no_grad():
        xhat, logJ_rt = cpl.atomic_backward(x_active=y.detach(), x_frozen=x_frozen, parity=parity,
                                            net=net, log0=logJ.detach())
    return dict(x_active=npy(x_active), out=npy(out), log0=npy(log0), y=npy(y), logJ=npy(logJ),
                grad_x=npy(gx), grad_out=npy(gout), xhat=npy(xhat), logJ_rt=npy(logJ_rt),
                parity=np.int64(parity), shape=np.array(shape, dtype=np.int64))


def gen_atoms():
    store = {}
    meta = []

    def add(tag, d, **kw):
        for k, v in d.items():
            store[f"{tag}/{k}"] = v
        meta.append(tag)

    seed = 1000
    shapes = {1: (8,), 2: (6, 4), 3: (4, 6, 4), 4: (4, 4, 2, 6)}
    for d, shape in shapes.items():
        for parity in (0, 1):
            seed += 1
            add(f"affine/d{d}p{parity}", atom_case(AffineCoupling_, shape, 4, 2, parity, seed))
            seed += 1
            add(f"shift/d{d}p{parity}", atom_case(ShiftCoupling_, shape, 4, 1, parity, seed))
            for m in (2, 4, 10, 16):
                seed += 1
                add(f"rqs_lin/d{d}p{parity}m{m}",
                    atom_case(RQSplineCoupling_, shape, 4, 3 * m - 2, parity, seed, xscale=1.6, oscale=1.2,
                              knot_hits=True, xlim=(-2.0, 2.0), ylim=(-2.5, 1.5),
                              extrap={'left': 'linear', 'right': 'linear'}))
            seed += 1
            add(f"rqs_anti/d{d}p{parity}m6",
                atom_case(RQSplineCoupling_, shape, 4, 16, parity, seed, xscale=1.0, oscale=1.0,
                          xlim=(0.0, 2.0), ylim=(0.0, 2.0), extrap={'left': 'anti', 'right': 'linear'}))
            seed += 1
            add(f"rqs_none/d{d}p{parity}m5",
                atom_case(RQSplineCoupling_, shape, 4, 13, parity, seed, oscale=1.0,
                          x_uniform=(0.02, 0.98), xlim=(0.0, 1.0), ylim=(0.0, 1.0), extrap={}))
        seed += 1
        add(f"rqs_onesided/d{d}",
            atom_case(RQSplineCoupling_, shape, 4, 10, 0, seed, oscale=1.0, x_uniform=(0.05, 3.0),
                      xlim=(0.0, 1.0), ylim=(0.0, 1.0), extrap={'right': 'linear'}))
        seed += 1
        add(f"multirqs/d{d}",
            atom_case(MultiRQSplineCoupling_, shape, 3, 2 * 10, 1, seed, xscale=1.5, oscale=1.0, x_channels=2,
                      xlims=[(-2.0, 2.0), (-1.0, 3.0)], ylims=[(-2.0, 2.0), (-3.0, 1.0)],
                      extraps=[{'left': 'linear', 'right': 'linear'}] * 2))
    # fixed knots_x (1-D) : out has 2m-1 channels
    seed += 1
    kx_fixed = torch.tensor([-1.0, -0.4, 0.1, 0.5, 1.0])
    d = atom_case(RQSplineCoupling_, (6, 4), 4, 9, 0, seed, xscale=0.9, oscale=1.0,
                  xlim=(-1.0, 1.0), ylim=(-1.0, 1.0), knots_x=kx_fixed,
                  extrap={'left': 'linear', 'right': 'linear'})
    d['knots_x'] = npy(kx_fixed)
    add("rqs_fixedx/d2", d)
    store['_cases'] = np.array(meta)
    save("atoms", **store)


# -------------------------------------------------------------------- DistConvertor_
def gen_distconv():
    store, meta = {}, []
    seed = 2000
    for symmetric in (False, True):
        for smooth in (False, True):
            for m, shape in ((10, (1,)), (4, (6, 4)), (16, (3, 4, 2))):
                seed += 1
                g = torch.Generator().manual_seed(seed)
                net_ = DistConvertor_(m, symmetric=symmetric, smooth=smooth)
                sp = net_.spline_layer_
                with torch.no_grad():
                    sp.weights_x.copy_(0.7 * torch.randn(m - 1, generator=g))
                    sp.weights_y.copy_(0.7 * torch.randn(m - 1, generator=g))
                    if not smooth:
                        sp.weights_d.copy_(0.7 * torch.randn(m, generator=g))
                x = (2.0 * torch.randn((5,) + shape, generator=g)).requires_grad_(True)
                log0 = 0.1 * torch.randn(5, generator=g)
                y, logJ = net_(x, log0)
                loss = logJ.mean() + (y ** 2).mean()
                params = [sp.weights_x, sp.weights_y] + ([] if smooth else [sp.weights_d])
                grads = torch.autograd.grad(loss, [x] + params)
                with torch.no_grad():
                    xhat, logJ_rt = net_.backward(y.detach(), logJ.detach())
                tag = f"dc/sym{int(symmetric)}sm{int(smooth)}m{m}"
                meta.append(tag)
                store.update({f"{tag}/x": npy(x), f"{tag}/log0": npy(log0), f"{tag}/y": npy(y),
                              f"{tag}/logJ": npy(logJ), f"{tag}/xhat": npy(xhat),
                              f"{tag}/logJ_rt": npy(logJ_rt),
                              f"{tag}/wx": npy(sp.weights_x), f"{tag}/wy": npy(sp.weights_y),
                              f"{tag}/grad_x": npy(grads[0]), f"{tag}/grad_wx": npy(grads[1]),
                              f"{tag}/grad_wy": npy(grads[2])})
                if not smooth:
                    store[f"{tag}/wd"] = npy(sp.weights_d)
                    store[f"{tag}/grad_wd"] = npy(grads[3])
    store['_cases'] = np.array(meta)
    save("distconv", **store)


# ------------------------------------------------------------------- whole Coupling_
def gen_blocks():
    store, meta = {}, []
    seed = 3000
    shapes = {1: (8,), 2: (4, 6), 3: (4, 4, 4), 4: (4, 2, 4, 4)}
    for d, shape in shapes.items():
        for kind in ("affine", "rqs"):
            seed += 1
            torch.manual_seed(seed)
            m = 6
            n_out = 2 if kind == "affine" else 3 * m - 2
            nets = [ConvAct(1, n_out, 3, conv_dim=d, hidden_sizes=[4, 4], acts=['tanh', 'tanh', None])
                    for _ in range(3)]
            mask = EvenOddMask(shape=shape)
            if kind == "affine":
                cpl = AffineCoupling_(nets, mask=mask)
            else:
                cpl = RQSplineCoupling_(nets, mask=mask, xlim=(-3.0, 3.0), ylim=(-3.0, 3.0),
                                        extrap={'left': 'linear', 'right': 'linear'})
            x = 1.3 * torch.randn((3,) + shape)
            x.requires_grad_(True)
            y, logJ = cpl(x)
            loss = logJ.mean() + (y ** 2).mean()
            plist = list(cpl.parameters())
            grads = torch.autograd.grad(loss, [x] + plist)
            with torch.no_grad():
                xhat, logJ_rt = cpl.backward(y.detach(), logJ.detach())
            tag = f"{kind}/d{d}"
            meta.append(tag)
            store.update({f"{tag}/x": npy(x), f"{tag}/y": npy(y), f"{tag}/logJ": npy(logJ),
                          f"{tag}/xhat": npy(xhat), f"{tag}/logJ_rt": npy(logJ_rt),
                          f"{tag}/grad_x": npy(grads[0]), f"{tag}/shape": np.array(shape)})
            names = [n for n, _ in cpl.named_parameters()]
            for n, p, gp in zip(names, plist, grads[1:]):
                store[f"{tag}/param/{n}"] = npy(p)
                store[f"{tag}/gparam/{n}"] = npy(gp)
            store[f"{tag}/mask"] = npy(mask._mask)
    store['_cases'] = np.array(meta)
    save("blocks", **store)


# ----------------------------------------------------------------------- end points
def gen_callers():
    store = {}
    # README model (config 1) with non-trivial spline weights: one Fitter.step-like tuple
    torch.manual_seed(4001)
    prior = NormalPrior(shape=(1,))
    action = ScalarPhi4Action(kappa=0, m_sq=-1.2, lambd=0.5)
    net_ = DistConvertor_(knots_len=10, symmetric=True)
    sp = net_.spline_layer_
    with torch.no_grad():
        sp.weights_x.copy_(0.5 * torch.randn(9))
        sp.weights_y.copy_(0.5 * torch.randn(9))
        sp.weights_d.copy_(0.5 * torch.randn(10))
    model = Model(net_=net_, prior=prior, action=action)
    x, logr = prior.sample_(128)
    y, logJ = net_(x)
    logq = logr - logJ
    logp = -action(y)
    loss = model.fit.calc_kl_mean(logq, logp)
    store.update({"c1/x": npy(x), "c1/logr": npy(logr), "c1/y": npy(y), "c1/logJ": npy(logJ),
                  "c1/logq": npy(logq), "c1/logp": npy(logp), "c1/loss": npy(loss),
                  "c1/wx": npy(sp.weights_x), "c1/wy": npy(sp.weights_y), "c1/wd": npy(sp.weights_d)})
    # log_prob path (net_.backward + prior.log_prob)
    with torch.no_grad():
        store["c1/log_prob"] = npy(model.posterior.log_prob(y.detach()))
    # actions and prior log-density on lattices
    g = torch.Generator().manual_seed(4002)
    for d, shape in {1: (8,), 2: (6, 4), 3: (4, 6, 4), 4: (4, 4, 2, 6)}.items():
        cfg = torch.randn((5,) + shape, generator=g)
        act = ScalarPhi4Action(kappa=0.67, m_sq=-4 * 0.67, lambd=0.5)
        store[f"phi4/d{d}/cfg"] = npy(cfg)
        store[f"phi4/d{d}/S"] = npy(act(cfg))
        store[f"phi4/d{d}/logr"] = npy(NormalPrior(shape=shape).log_prob(cfg))
    store["phi4/coef"] = np.array([0.67, -4 * 0.67, 0.5])
    # masks
    for shape in ((4,), (4, 4), (3, 5), (2, 4, 6), (2, 2, 4, 2)):
        for parity in (0, 1):
            key = "mask/" + "x".join(map(str, shape)) + f"/p{parity}"
            store[key] = npy(EvenOddMask(shape=shape, parity=parity)._mask)
    save("callers", **store)


if __name__ == "__main__":
    gen_atoms()
    gen_distconv()
    gen_blocks()
    gen_callers()
