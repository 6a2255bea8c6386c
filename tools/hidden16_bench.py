"""A spline layer whose net has hidden width 16 (1-16-16-46) on 32^4: which kernels take it and how fast.  python tools/hidden16_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")
import torch
from normflow__amd.mask import EvenOddMask
from normflow__amd.nn import ConvAct, RQSplineCoupling_, ModuleList_

dev = torch.device("cuda", 0)
lattice = (32, 32, 32, 32)
for hidden in (8, 16):
    torch.manual_seed(0)
    nets = [ConvAct(1, 46, 3, conv_dim=4, hidden_sizes=[hidden, hidden], acts=['tanh', 'tanh', None]) for _ in range(2)]
    cpl = RQSplineCoupling_(nets, mask=EvenOddMask(shape=lattice), xlim=(-5.0, 5.0), ylim=(-5.0, 5.0), extrap={'left': 'linear', 'right': 'linear'})
    net_ = ModuleList_([cpl])
    net_.to(device=dev, dtype=torch.float32)
    B = 64
    x = torch.randn((B,) + lattice, device=dev, dtype=torch.float32)
    with torch.no_grad():
        net_(x); net_(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            y, lj = net_(x)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"hidden {hidden}: {dt * 1e3 / 2:8.2f} ms per layer at B = {B}  ({dt * 1e3 / 2 * 256 / B:7.1f} ms per 256 samples)", flush=True)
