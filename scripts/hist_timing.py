"""Phase timing of the counting histogram kernels (needs a -DPCO_HIST_TIMING build: PCO_GFX_LIB=ab/libpco_gfx_histtiming.so)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gpu_util as U
from pcodec_amd import _lib as G
L = G.lib()
names = ["zero + count", "prefix", "rank lookups", "emit (parallel or state machine)"]
def run(tag, arrays, kw):
    z = (C.c_ulonglong * 16)()
    L.pco_gfx_debug_hist_timing(z, 1)
    U.gpu_batched(arrays, G.make_config(**kw))
    L.pco_gfx_debug_hist_timing(z, 0)
    v = list(z)
    for base, kn in ((0, "enc_hist_kernel"), (8, "enc_hist_wide_kernel")):
        n = max(v[base + 4], 1)
        print(tag, kn, "vars", v[base + 4], " ".join(f"{names[k]}: {v[base + k] / n:.0f}" for k in range(4)))
run("c2 u64 ramp delta-1", [U.synth("c2", seed=s) for s in range(512)], dict(mode=1, delta=2, delta_order=1))
run("c3 f64 decimals float-mult", [U.synth("c3", seed=s) for s in range(512)], dict(mode=2, mode_f64=0.01, delta=1))
