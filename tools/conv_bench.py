#!/usr/bin/env python3
"""Micro-benchmark of the K5 conv kernel on the three layer shapes of the bench network
(32^4 lattice): time per launch (HIP events) and useful TFLOP/s.  NF_CONV_DBG=1 skips the
LDS staging, =2 skips the MFMA loop (profiling ablation only; results are then wrong)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")
import torch
from normflow__amd import _hip

dev = torch.device("cuda", 0)
lat = tuple(int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "32,32,32,32").split(","))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
V = 1
for n in lat:
    V *= n
for name, cin, cout, act, compact in (("L1 1->8 tanh", 1, 8, 1, False), ("L2 8->8 tanh", 8, 8, 1, False),
                                      ("L3 8->46 compact", 8, 46, 0, True), ("L3 8->46 full", 8, 46, 0, False)):
    x = torch.randn((B, cin) + lat, device=dev)
    w = 0.1 * torch.randn((cout, cin) + (3,) * len(lat), device=dev)
    b = torch.randn(cout, device=dev)
    f = lambda: _hip._conv_launch(x, w, b, act, compact, 0)
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    reps = 5
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    macs = B * V * (0.5 if compact else 1.0) * cin * cout * 3 ** len(lat)
    print(f"{name:18s} B={B} {ms:8.3f} ms  {2*macs/ms/1e9:7.2f} TFLOP/s useful  dbg={os.environ.get('NF_CONV_DBG','0')}")
