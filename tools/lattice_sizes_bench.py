import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tools'))
os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")
import config_bench as cb
for L, B in ((16, 4096), (32, 1024), (64, 256)):
    cb.run(f"2-D {L}x{L}, 8 rqs", (L, L), ["rqs"] * 8, B, reps=5)
    cb.run_graphed(f"2-D {L}x{L}, 8 rqs (graph)", (L, L), ["rqs"] * 8, B)
for L, B in ((8, 4096), (32, 32)):
    cb.run(f"3-D {L}^3, 8 rqs", (L, L, L), ["rqs"] * 8, B, reps=5)
