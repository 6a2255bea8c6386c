"""Same-box A/B of the packed-weight cache (normflow__amd._hip._cached_pack): the bench's timed loop with and without it.
    python tools/pack_cache_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")
import torch
import bench
from normflow__amd import _hip

dev = torch.device("cuda", 0)
lattice = (32, 32, 32, 32)
net_, cpl = bench.build_net(lattice, 8, 16, dev, seed=2024)
x = torch.randn((1024,) + lattice, device=dev, dtype=torch.float32)


def timed(steps=3):
    with torch.no_grad():
        net_(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            net_(x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


orig = _hip._cached_pack
for rnd in range(2):
    _hip._cached_pack = orig
    a = timed()
    _hip._cached_pack = lambda w, kind, fn: fn(w.detach())
    b = timed()
    print(f"round {rnd}: cached {1e3 * a:.1f} ms/step = {1024 / a:.1f} configs/s | repacked every call {1e3 * b:.1f} ms/step = {1024 / b:.1f} configs/s")
