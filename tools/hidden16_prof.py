"""One spline layer with hidden width 16 on 32^4, batch 64 -- under rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")
import torch
from normflow__amd.mask import EvenOddMask
from normflow__amd.nn import ConvAct, RQSplineCoupling_, ModuleList_
dev = torch.device("cuda", 0)
lattice = (32, 32, 32, 32)
torch.manual_seed(0)
nets = [ConvAct(1, 46, 3, conv_dim=4, hidden_sizes=[16, 16], acts=['tanh', 'tanh', None]) for _ in range(2)]
cpl = RQSplineCoupling_(nets, mask=EvenOddMask(shape=lattice), xlim=(-5.0, 5.0), ylim=(-5.0, 5.0), extrap={'left': 'linear', 'right': 'linear'})
net_ = ModuleList_([cpl]); net_.to(device=dev, dtype=torch.float32)
x = torch.randn((64,) + lattice, device=dev, dtype=torch.float32)
with torch.no_grad():
    for _ in range(3):
        net_(x)
torch.cuda.synchronize()
