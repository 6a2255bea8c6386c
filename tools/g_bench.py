"""Time the split-fp16 hidden layer (nf_conv_fwd_split16) and the first layer alone at the bench's slab shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")
import torch
from normflow__amd import _hip
if os.environ.get('NF_LIB'):              # an ablation / instrumented build of the library (tools/_build/)
    _hip.LIB_PATH = os.environ['NF_LIB']
DEV = torch.device("cuda:0")
lat, slab = (32,) * 4, int(os.environ.get("SLAB", 64))
V = 32 ** 4
g = torch.Generator(device=DEV).manual_seed(1)
h16 = (torch.rand((slab, V, 16), device=DEV, generator=g) - 0.5).half()
w = 0.1 * torch.randn((8, 8, 3, 3, 3, 3), device=DEV, generator=g)
b = 0.1 * torch.randn(8, device=DEV, generator=g)
if os.environ.get('ZERO') == '1':      # zero operands: same instruction stream, least switching power
    h16.zero_(); w.zero_()
def timeit(f, n=3):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = timeit(lambda: _hip.conv_layer_split16(h16, w, b, 1, lat))
print(f"K5g 8->8 split16   slab {slab}: {ms:8.3f} ms   {slab * V * 64 / ms / 1e6:7.1f} GB/s (32 B in + 32 B out per site)  dbg={os.environ.get('NF_CONVG_DBG','0')}")
x = torch.randn((slab, 1) + lat, device=DEV, generator=g)
w1 = 0.3 * torch.randn((8, 1, 3, 3, 3, 3), device=DEV, generator=g)
ms = timeit(lambda: _hip.conv_layer(x, w1, b, 1, compact=2))
print(f"K5c 1->8 -> split16 slab {slab}: {ms:8.3f} ms   {slab * V * 36 / ms / 1e6:7.1f} GB/s (4 B in + 32 B out per site)")
