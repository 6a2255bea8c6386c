"""Import side effects kept from the reference (src/device/__init__.py:7-13): the
default dtype becomes float64 and the default device 'cuda' when a GPU is present, so
that user scripts written against the reference behave identically.  Set
NORMFLOW_AMD_KEEP_TORCH_DEFAULTS=1 to opt out."""
import os

import torch

from ._core import ModelDeviceHandler

torch_device = 'cuda' if torch.cuda.is_available() else 'cpu'

if os.environ.get("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "0") != "1":
    torch.set_default_device(torch_device)
    torch.set_default_dtype(torch.float64)
