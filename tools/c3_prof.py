"""One config-3 step for rocprofv3 (counters / kernel trace): python tools/c3_prof.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import config_bench as cb
cb.run("c3 16^3, 8 rqs m=16 (K5s)", (16, 16, 16), ["rqs"] * 8, 1024, reps=2)
