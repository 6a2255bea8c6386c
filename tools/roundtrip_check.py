"""Round trip x -> y -> x' of the headline network on a few samples under the two product arithmetics: how large is the
worst site, how many sites are off, and is it the arithmetic or the conditioning?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")
import torch
import bench
from normflow__amd import _hip

dev = torch.device("cuda", 0)
lattice = (32, 32, 32, 32)
net_, cpl = bench.build_net(lattice, 8, 16, dev, seed=2024)
x = torch.randn((16,) + lattice, device=dev, dtype=torch.float32)
with torch.no_grad():
    for name, split in (("split-fp16 products", True), ("fp32 products", False)):
        with _hip.options(split16=split):
            y, lj = net_(x)
            xb, lb = net_.backward(y)
        e = (xb - x).abs().flatten()
        print(f"{name}: max {float(e.max()):.3e}  sites > 1e-3: {int((e > 1e-3).sum())}  > 1e-4: {int((e > 1e-4).sum())} of {e.numel()}  "
              f"median {float(e.median()):.2e}  logJ rel {float(((lj + lb).abs() / lj.abs()).max()):.2e}", flush=True)
    # one layer only
    one, cpl1 = bench.build_net(lattice, 1, 16, dev, seed=2024)
    y, lj = one(x)
    xb, lb = one.backward(y)
    e = (xb - x).abs().flatten()
    print(f"one layer: max {float(e.max()):.3e}  sites > 1e-4: {int((e > 1e-4).sum())}  logJ rel {float(((lj + lb).abs() / lj.abs()).max()):.2e}")
    i = int(e.argmax())
    print("worst site: x", float(x.flatten()[i]), "y", float(y.flatten()[i]), "x'", float(xb.flatten()[i]))
