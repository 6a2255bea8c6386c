"""Compare the first-layer kernel (K5c) against the fp64 definition on a few shapes; NF_CONV_C1_BF16=0/1 selects the arithmetic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")
import torch
from normflow__amd import _hip
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
for lat, B in (((2, 2, 4, 32), 3), ((4, 2, 6, 32), 5), ((8, 8, 8, 32), 2), ((32, 32, 32, 32), 2)):
    x = torch.randn((B, 1) + lat, device=dev, generator=g)
    w = 0.3 * torch.randn((8, 1, 3, 3, 3, 3), device=dev, generator=g)
    b = 0.1 * torch.randn(8, device=dev, generator=g)
    y = _hip.conv_layer(x, w, b, 0, compact=0)                 # act 0 = identity
    path = _hip.load().nf_conv_last_path()
    # fp64 definition: circular correlation
    xd, wd = x.double(), w.double()
    ref = torch.zeros((B, 8) + lat, dtype=torch.float64, device=dev)
    for j0 in range(3):
        for j1 in range(3):
            for j2 in range(3):
                for j3 in range(3):
                    sh = torch.roll(xd, shifts=(1 - j0, 1 - j1, 1 - j2, 1 - j3), dims=(2, 3, 4, 5))
                    ref += sh * wd[None, :, 0, j0, j1, j2, j3, None, None, None, None]
    ref += b.double()[None, :, None, None, None, None]
    err = (y.double() - ref).abs().max().item()
    print(f"lattice {lat} B {B}: path {path}  max abs err {err:.3e}  (|ref| max {ref.abs().max().item():.2f})  finite {bool(torch.isfinite(y).all())}")
