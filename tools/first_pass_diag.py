"""Per-pass wall time of the headline step in a fresh process (the first passes of the first process on a box are slow: why?).
python tools/first_pass_diag.py"""
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
y = lj = None
with torch.no_grad():
    for i in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y, lj = net_(x)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = torch.cuda.memory_stats(dev)
        print(f"pass {i}: {dt * 1e3:8.1f} ms  reserved {st['reserved_bytes.all.current'] / 2**30:6.1f} GiB  allocated {st['allocated_bytes.all.current'] / 2**30:6.1f} GiB  "
              f"segments {st['segment.all.current']}  cudaMalloc calls {st['num_device_alloc']}  retries {st['num_alloc_retries']}", flush=True)
