"""Time the three kernels of one coupling layer of the bench network at the pipeline's slab (HIP events), one line each.
    python tools/kbench.py [--batch 256] [--reps 5]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--lattice", type=str, default="32,32,32,32")
ap.add_argument("--tag", type=str, default="")
ap.add_argument("--only", type=str, default="")
a = ap.parse_args()
dev = torch.device("cuda", 0)
lattice = tuple(int(s) for s in a.lattice.split(","))
net_, cpl = bench.build_net(lattice, 2, 16, dev, seed=2024)
if a.only == "k2":
    kt = bench.time_rqs_kernel(cpl, lattice, 16, dev, a.reps, True)
    print(f"[kbench {a.tag}] K2 slab {kt['slab']} {1e3*kt['seconds']:.3f} ms  {kt['gbs']:.0f} GB/s", flush=True)
    sys.exit(0)
if a.only in ("g", "c"):
    oth = bench.time_hidden_layers(cpl, lattice, dev, a.reps, a.batch)
    print(f"[kbench {a.tag}] K5g {oth[1]['launch_ms']:.3f} ms  K5c {oth[0]['launch_ms']:.3f} ms", flush=True)
    sys.exit(0)
ft = bench.time_fused_last_layer(cpl, lattice, 16, dev, a.reps, a.batch)
if a.only == "h":
    print(f"[kbench {a.tag}] K5h {1e3*ft['seconds']:.3f} ms", flush=True)
    sys.exit(0)
oth = bench.time_hidden_layers(cpl, lattice, dev, a.reps, a.batch)
print(f"[kbench {a.tag}] slab {ft['slab']}  K5h {1e3*ft['seconds']:.3f} ms  K5g {oth[1]['launch_ms']:.3f} ms  K5c {oth[0]['launch_ms']:.3f} ms  "
      f"sum {1e3*ft['seconds'] + oth[1]['launch_ms'] + oth[0]['launch_ms']:.3f} ms", flush=True)
