"""The headline step (32^4, 8 spline layers, batch 1024) at different slab sizes of the fused atoms: fewer, longer launches of each
kernel against more hidden activations in flight.  python tools/slab_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")
import torch
import bench
from normflow__amd.nn.scalar import couplings_

dev = torch.device("cuda", 0)
lattice = (32, 32, 32, 32)
net_, cpl = bench.build_net(lattice, 8, 16, dev, seed=2024)
x = torch.randn((1024,) + lattice, device=dev, dtype=torch.float32)
for rep in range(2):
    for gib in (8, 16, 32, 8):
        couplings_.set_slab_bytes(hidden=gib << 30)
        with torch.no_grad():
            net_(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                y, lj = net_(x)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"hidden slab {gib:3d} GiB = {gib * 32} samples: {dt * 1e3:8.1f} ms/step  {1024 / dt:7.1f} configs/s  logJ[0] {float(lj[0]):.6f}", flush=True)
