"""Phase timing of enc_lookback_kernel (needs a -DPCO_LB_TIMING build)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import gpu_util as U
from pcodec_amd import _lib as G
L = G.lib()
names = ["P1 load + hash reads", "P1 hazard loop", "P1 atomics, proposals, statics", "rounds", "store + apply"]
def run(tag, arrays, kw):
    z = (C.c_ulonglong * 16)()
    L.pco_gfx_debug_lb_timing(z, 1)
    U.gpu_batched(arrays, G.make_config(**kw))
    L.pco_gfx_debug_lb_timing(z, 0)
    v = list(z); tiles = max(v[6], 1)
    print(tag, "pages", v[7], "tiles/page", v[6] / max(v[7], 1), "rounds/tile", v[5] / tiles)
    for k in range(5): print(f"   {names[k]:34s} {v[k] / tiles:10.0f} cycles per tile")
rng = np.random.default_rng(1)
run("i64 seasonal x1024", [U.synth("c4", seed=s) for s in range(1024)], dict(mode=1, delta=3))
run("i64 seasonal x64", [U.synth("c4", seed=s) for s in range(64)], dict(mode=1, delta=3))
run("u32 random x256", [rng.integers(0, 1 << 32, 1 << 18, dtype=np.uint64).astype(np.uint32) for _ in range(256)], dict(mode=1, delta=3))
run("u64 ramp x256", [U.synth("c2", seed=s) for s in range(256)], dict(mode=1, delta=3))
run("trial-sized: 2048 x u64 random ints 1000..10000 n=6563", [rng.integers(1000, 10000, 6563).astype(np.uint64) for _ in range(2048)], dict(mode=1, delta=3))
run("trial-sized: 2048 x u64 noisy ramp n=6563", [U.synth("c2", seed=s)[:6563] for s in range(2048)], dict(mode=1, delta=3))
