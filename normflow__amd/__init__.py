"""normflow__amd: MI355X-native coupling-layer hot path behind the normflow API.

    from normflow__amd import Model
    from normflow__amd.nn import RQSplineCoupling_, AffineCoupling_, DistConvertor_, ConvAct
    from normflow__amd.mask import EvenOddMask
"""
from ._normflowcore import Model, np, torch
from ._normflowcore import backward_sanitychecker

from . import action
from . import device
from . import mask
from . import nn
from . import prior
from . import mcmc

__version__ = "0.1.0"
from .graphs import GraphedFlow, GraphedTrainStep
