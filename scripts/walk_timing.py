"""Cycle breakdown of dec_walk_kernel's rounds (block 0): needs a library built with -DPCO_WALK_TIMING
(scripts/build_variant.sh timing -DPCO_WALK_TIMING) and PCO_GFX_LIB pointing at it.  usage: walk_timing.py [chunks] [workload]"""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
chunks = sys.argv[1] if len(sys.argv) > 1 else "2048"
wl = sys.argv[2] if len(sys.argv) > 2 else "c2"
sys.argv = ["bench.py", "--steps", "1", "--warmup", "1", "--chunks", chunks, "--workload", wl, "--no-cpu-baseline"]
import bench
bench.main()
from pcodec_amd import _lib as G
buf = (C.c_ulonglong * 8)()
G.lib().pco_gfx_debug_walk_timing(buf)
stage, walk, tail, rounds, t0, t1 = list(buf)[:6]
print(f"rounds {rounds}  per round: stage {stage / rounds:.0f}  walk {walk / rounds:.0f}  tail {tail / rounds:.0f}  (clock ticks); kernel total {t1 - t0}")
