import os, sys
sys.path.insert(0, '/root/repo'); os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")
import torch
from normflow__amd import _hip
dev = torch.device("cuda", 0)
lat = (32, 32, 32, 32); B = 16
for cin, cout, act, compact in ((8, 8, 1, False), (8, 46, 0, True)):
    x = torch.randn((B, cin) + lat, device=dev); w = 0.1 * torch.randn((cout, cin) + (3,) * 4, device=dev); b = torch.randn(cout, device=dev)
    for _ in range(2):
        _hip._conv_launch(x, w, b, act, compact, 0)
    torch.cuda.synchronize()
