import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")
import numpy as np, torch
from normflow__amd import _hip
from oracle import nf_oracle as O
DEV = torch.device("cuda", 0)
for shape, B, xs in (((2, 2, 2, 32), 1, 1.0), ((2, 4, 4, 32), 2, 1.0), ((4, 2, 6, 32), 3, 30.0), ((4, 4, 8, 32), 9, 1e-3), ((8, 8, 8, 32), 3, 1.0), ((16, 16, 16, 32), 2, 1.0), ((2, 2, 4, 48), 2, 1.0), ((4, 4, 4, 64), 3, 1.0), ((4, 2, 6, 48), 5, 1.0), ((2, 2, 2, 96), 1, 1.0)):
    g = torch.Generator(device='cpu').manual_seed(5)
    x = xs * torch.randn((B, 1) + shape, generator=g, dtype=torch.float64, device='cpu')
    w = 0.3 * torch.randn((8, 1, 3, 3, 3, 3), generator=g, dtype=torch.float64, device='cpu') / max(1.0, xs)
    b = 0.3 * torch.randn(8, generator=g, dtype=torch.float64, device='cpu')
    ref = torch.tanh(O.circular_conv_fast(x, w, b))
    out16 = _hip.conv_layer(x.to(DEV, torch.float32), w.to(DEV, torch.float32), b.to(DEV, torch.float32), _hip.ACT_CODES['tanh'], compact=2)
    torch.cuda.synchronize()
    out = _hip.from_split16(out16, shape).double().cpu()
    err = (out - ref).abs()
    print(shape, B, xs, "max err", float(err.max()), flush=True)
    if err.max() > 1e-4:
        bad = (err > 1e-4)
        print("  bad fraction", float(bad.float().mean()))
        for ax, name in enumerate(["b", "c", "x0", "x1", "x2", "x3"]):
            dims = [d for d in range(6) if d != ax]
            print("  bad fraction along", name, [round(float(v), 3) for v in bad.float().mean(dim=dims)][:34])
