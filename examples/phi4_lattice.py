#!/usr/bin/env python3
"""Train a phi^4 flow on an MI355X with this package -- the network of the reference's
examples/scalar_affine.py (spectral block, DistConvertor_, affine couplings with ConvAct nets,
DistConvertor_), or a stack of RQ-spline couplings, assembled from the same names.

    python examples/phi4_lattice.py --lat 8,8 --epochs 500
    python examples/phi4_lattice.py --lat 16,16,16 --kind rqs --layers 4 --epochs 100
    torchrun-free data parallel:  --nranks 8   (device_handler.spawnprocesses, one process per GPU)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a source checkout

import normflow__amd as nf          # the reference:  import normflow as nf
from normflow__amd.action import ScalarPhi4Action
from normflow__amd.mask import EvenOddMask
from normflow__amd.nn import (AffineCoupling_, ConvAct, DistConvertor_, FFTNet_, MeanFieldNet_, ModuleList_,
                              PSDBlock_, RQSplineCoupling_)
from normflow__amd.prior import NormalPrior


def build_net(lat, kind, layers, knots):
    d = len(lat)
    mask = EvenOddMask(shape=lat)

    def param_net(channels):
        return ConvAct(in_channels=1, out_channels=channels, hidden_sizes=[8, 8], kernel_size=3, conv_dim=d,
                       acts=('tanh', 'tanh', None), bias=(kind == 'rqs'))

    if kind == 'affine':
        return ModuleList_([
            PSDBlock_(mfnet_=MeanFieldNet_.build(knots_len=10, symmetric=True, final_scale=True, smooth=True),
                      fftnet_=FFTNet_.build(lat, knots_len=10, ignore_zeromode=True)),
            DistConvertor_(50, symmetric=True, smooth=True),
            AffineCoupling_([param_net(2) for _ in range(layers)], mask=mask),
            DistConvertor_(50, symmetric=True, smooth=True)])
    return ModuleList_([
        RQSplineCoupling_([param_net(3 * knots - 2) for _ in range(layers)], mask=mask, xlim=(-5, 5), ylim=(-5, 5),
                          extrap={'left': 'linear', 'right': 'linear'})])


def fit(model, **kw):
    model.fit(**kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lat", default="8,8")
    ap.add_argument("--kind", choices=("affine", "rqs"), default="affine")
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--knots", type=int, default=16)
    ap.add_argument("--epochs", type=int, default=500)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--nranks", type=int, default=1)
    ap.add_argument("--kappa", type=float, default=0.67)
    ap.add_argument("--m_sq", type=float, default=-4 * 0.67)
    ap.add_argument("--lambd", type=float, default=0.5)
    a = ap.parse_args()
    lat = tuple(int(n) for n in a.lat.split(","))
    model = nf.Model(net_=build_net(lat, a.kind, a.layers, a.knots), prior=NormalPrior(shape=lat),
                     action=ScalarPhi4Action(kappa=a.kappa, m_sq=a.m_sq, lambd=a.lambd))
    print("number of model parameters =", model.net_.npar)
    kw = dict(n_epochs=a.epochs, batch_size=a.batch // a.nranks, checkpoint_dict=dict(print_stride=max(1, a.epochs // 10)))
    if a.nranks > 1:
        model.device_handler.spawnprocesses(fit, a.nranks, **kw)
    else:
        model.fit(**kw)
        nf.backward_sanitychecker(model)


if __name__ == "__main__":
    main()
