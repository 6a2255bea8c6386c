import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")
import numpy as np, torch
from normflow__amd import _hip
from oracle import nf_oracle as O
DEV = torch.device("cuda", 0)
for shape, B in (((2, 2, 2, 32), 1), ((2, 2, 4, 32), 1), ((4, 2, 6, 32), 3), ((4, 4, 8, 32), 9), ((2, 2, 4, 48), 2), ((4, 4, 4, 64), 3), ((4, 2, 6, 48), 5), ((2, 2, 2, 96), 1), ((8, 8, 8, 48), 2)):
    g = torch.Generator(device='cpu').manual_seed(5)
    h = torch.tanh(torch.randn((B, 8) + shape, generator=g, dtype=torch.float64, device='cpu'))
    w = 0.2 * torch.randn((8, 8, 3, 3, 3, 3), generator=g, dtype=torch.float64, device='cpu')
    b = 0.3 * torch.randn(8, generator=g, dtype=torch.float64, device='cpu')
    ref = torch.tanh(O.circular_conv_fast(h, w, b))
    h16 = _hip.to_split16(h.to(DEV, torch.float32))
    out16 = _hip.conv_layer_split16(h16, w.to(DEV, torch.float32), b.to(DEV, torch.float32), _hip.ACT_CODES['tanh'], shape)
    torch.cuda.synchronize()
    out = _hip.from_split16(out16, shape).double().cpu()
    err = (out - ref).abs()
    print(shape, B, "max err", float(err.max()))
    if err.max() > 1e-4:
        bad = (err > 1e-4)
        print("  bad fraction", float(bad.float().mean()))
        for ax, name in enumerate(["b", "c", "x0", "x1", "x2", "x3"]):
            dims = [d for d in range(6) if d != ax]
            print("  bad fraction along", name, [round(float(v), 3) for v in bad.float().mean(dim=dims)][:34])
