import sys, torch
sys.path.insert(0, '.')
import os
os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")
from normflow__amd import _hip
from oracle import nf_oracle as O
for lattice, cin, cout, zero in (((16, 16), 8, 2, None), ((16, 16), 8, 2, 'hi'), ((16, 16), 8, 2, 'lo'), ((8, 8, 8, 8), 8, 46, None), ((8, 8, 8, 16), 8, 8, None), ((4,4,16, 16), 8, 2, None)):
    g = torch.Generator(device='cpu').manual_seed(1)
    d = len(lattice)
    x = torch.randn((3, cin) + lattice, generator=g, dtype=torch.float64)
    w = 0.3 * torch.randn((cout, cin) + (3,) * d, generator=g, dtype=torch.float64)
    if zero == 'hi': w[:, 4:] = 0
    if zero == 'lo': w[:, :4] = 0
    b = torch.randn(cout, generator=g, dtype=torch.float64)
    ref = O.circular_conv_fast(x, w, b)
    out = _hip.conv_layer(x.cuda().float(), w.cuda().float(), b.cuda().float(), 0)
    path = _hip.load().nf_conv_last_path()
    err = (out.double().cpu() - ref).abs()
    print(lattice, cin, cout, zero, 'path', path, 'max err', float(err.max()))
    if float(err.max()) > 1e-4:
        e2 = err.amax(dim=(0, 1))
        if d == 2:
            for r in range(lattice[0]):
                print('   ', ''.join('X' if v > 1e-4 else '.' for v in e2[r].tolist()))
