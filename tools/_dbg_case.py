import sys, torch
sys.path.insert(0, '.')
import os
os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")
from normflow__amd import _hip
from oracle import nf_oracle as O
for lattice, cout, B in (((8, 8, 8, 32), 8, 2), ((16, 16), 8, 3), ((4,4,4,8), 8, 2)):
    g = torch.Generator(device='cpu').manual_seed(1)
    d = len(lattice)
    x = torch.randn((B, 1) + lattice, generator=g, dtype=torch.float64)
    w = 0.3 * torch.randn((cout, 1) + (3,) * d, generator=g, dtype=torch.float64)
    b = torch.randn(cout, generator=g, dtype=torch.float64)
    ref = O.circular_conv_fast(x, w, b)
    out = _hip.conv_layer(x.cuda().float(), w.cuda().float(), b.cuda().float(), 0)
    path = _hip.load().nf_conv_last_path()
    err = (out.double().cpu() - ref).abs()
    print(lattice, cout, 'path', path, 'max err', float(err.max()), 'frac bad', float((err > 1e-4).double().mean()))
    if float(err.max()) > 1e-4:
        bad = (err > 1e-4)
        print('  per-channel bad', bad.double().mean(dim=(0,) + tuple(range(2, 2 + d))).tolist())
        print('  per-x3 bad', [round(v, 2) for v in bad.double().mean(dim=tuple(range(0, 1 + d))).tolist()])
        if d == 2:
            e2 = err.amax(dim=(0, 1))
            for r in range(lattice[0]):
                print('   ', ''.join('X' if v > 1e-4 else '.' for v in e2[r].tolist()))
