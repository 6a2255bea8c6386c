"""Forward against inverse pass of the headline network (32^4, 8 spline layers, batch 1024): the inverse is what
posterior.log_prob and backward_sanitychecker run.  python tools/inverse_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")
import torch
import bench

dev = torch.device("cuda", 0)
lattice = (32, 32, 32, 32)
net_, cpl = bench.build_net(lattice, 8, 16, dev, seed=2024)
x = torch.randn((1024,) + lattice, device=dev, dtype=torch.float32)
with torch.no_grad():
    y, lj = net_(x)
    for name, fn, arg in (("forward", net_.forward, x), ("inverse", net_.backward, y)):
        fn(arg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out, l = fn(arg)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"{name}: {dt * 1e3:8.1f} ms/step  {1024 / dt:7.1f} configs/s", flush=True)
    xb, lb = net_.backward(y)
    print("round trip max |x - x'|", float((xb - x).abs().max()), " max |logJ + logJ'|", float((lj + lb).abs().max() / lj.abs().max()))
