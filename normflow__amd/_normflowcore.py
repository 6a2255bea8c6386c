"""Compatibility shim: the reference keeps Model / Posterior / Fitter in one module of this
name; here they live in `model.py` and `fitter.py`."""
import numpy as np  # noqa: F401  (re-exported like the reference does)
import torch  # noqa: F401

from .fitter import Fitter  # noqa: F401
from .model import Model, Posterior, backward_sanitychecker  # noqa: F401
