"""Time the fused last layer (nf_conv_rqs) alone at the bench's slab shape.  NF_CONV_DBG / NF_CONV_SPLIT16 apply."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")
import torch
from normflow__amd import _hip
if os.environ.get('NF_LIB'):              # an ablation build of the library (tools/_build/)
    _hip.LIB_PATH = os.environ['NF_LIB']
DEV = torch.device("cuda:0")
lat, m, slab = (32,) * 4, 16, int(os.environ.get("SLAB", 64))
V = 32 ** 4
g = torch.Generator(device=DEV).manual_seed(1)
h = torch.tanh(torch.randn((slab, 8) + lat, device=DEV, generator=g))
x = torch.randn(slab, V, device=DEV, generator=g)
w = 0.1 * torch.randn((46, 8, 3, 3, 3, 3), device=DEV, generator=g)
b = 0.1 * torch.randn(46, device=DEV, generator=g)
if os.environ.get('ZERO') == '1':      # zero operands: same instruction stream, least switching power
    h.zero_(); w.zero_()
opts = _hip.make_rqs_opts(m, (-5.0, 5.0), (-5.0, 5.0), {'left': 'linear', 'right': 'linear'}, _hip.LAYOUT_PAIR)
if os.environ.get("SPLITIN", "1") == "1":          # what the pipeline feeds the kernel: (B, V, 16) halfs, hi | lo per site
    hp = h.reshape(slab, 8, V).permute(0, 2, 1).contiguous()
    hi = hp.half()
    h16 = torch.cat((hi, (hp - hi.float()).half()), dim=2).contiguous()
    del hp, hi
    f = lambda: _hip.conv_rqs(h16, w, b, x, None, 0, opts, False, unit_input=True, lattice=lat)
else:
    f = lambda: _hip.conv_rqs(h, w, b, x, None, 0, opts, False, unit_input=True)
for _ in range(2): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): f()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
flops = 2.0 * 81 * 8 * 46 * (V // 2) * slab
print(f"path {_hip.load().nf_conv_last_path()}  slab {slab}: {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s fp32-equivalent  dbg={os.environ.get('NF_CONV_DBG','0')}")
