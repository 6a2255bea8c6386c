"""BASELINE config 3 (16^3, 8 RQ-spline layers, batch 1024) on the small-lattice fused kernel (nf_conv_s.hip) and, for
comparison, on the fp32 kernels.  python tools/c3_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import config_bench as cb
from normflow__amd import _hip

if __name__ == "__main__":
    cb.run("c3 16^3, 8 rqs m=16 (K5s)", (16, 16, 16), ["rqs"] * 8, 1024, reps=10)
    cb.run("c3 16^3, 8 rqs m=16 (K5s)", (16, 16, 16), ["rqs"] * 8, 4096, reps=5)
    cb.run("c3 16^3, 8 rqs m=8  (K5s)", (16, 16, 16), ["rqs"] * 8, 1024, reps=10, m=8)
    with _hip.options(small8=False):
        cb.run("c3 16^3, m=16 (K5s, 4-wave form)", (16, 16, 16), ["rqs"] * 8, 1024, reps=10)
    with _hip.options(split16=False):
        cb.run("c3 16^3, fp32 kernels", (16, 16, 16), ["rqs"] * 8, 1024, reps=5)
