"""Phase timing of enc_hist_select_kernel (needs a -DPCO_SEL_TIMING build: PCO_GFX_LIB=ab/libpco_gfx_seltiming.so)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import gpu_util as U
from pcodec_amd import _lib as G
L = G.lib()
names = ["A sample+sort", "B,C segs+count", "D,E,F prefix/mark/list", "G gather", "H windows", "H2 block sorts", "I queries", "state machine"]
def run(tag, arrays, kw):
    z = (C.c_ulonglong * 16)()
    L.pco_gfx_debug_sel_timing(z, 1)
    U.gpu_batched(arrays, G.make_config(**kw))
    L.pco_gfx_debug_sel_timing(z, 0)
    v = list(z); n = max(v[8], 1)
    print(tag, "vars", v[8], "avg n_need", v[9] / n, "n_sub", v[10] / n, "n_big", v[11] / n, "n_seg", v[12] / n)
    tot = sum(v[:8])
    for k in range(8): print(f"   {names[k]:28s} {v[k] / n:12.0f} cycles  {100.0 * v[k] / max(tot, 1):5.1f}%")
rng = np.random.default_rng(1)
nch = 512
run("u32 uniform random, no delta", [rng.integers(0, 1 << 32, 1 << 18, dtype=np.uint64).astype(np.uint32) for _ in range(nch)], dict(mode=1, delta=1))
run("f32 normal, delta 1", [rng.standard_normal(1 << 18).astype(np.float32) for _ in range(nch)], dict(mode=1, delta=2, delta_order=1))
run("i32 lomax, delta 1", [(rng.pareto(0.5, 1 << 18) * 10).clip(0, 2e9).astype(np.int32) for _ in range(nch)], dict(mode=1, delta=2, delta_order=1))
run("i64 seasonal lookback", [U.synth("c4", seed=s) for s in range(64)], dict(mode=1, delta=3))

run("small: 2048 x u64 uniform n=6600 (trial-sized)", [rng.integers(0, 1 << 62, 6600, dtype=np.uint64) for _ in range(2048)], dict(mode=1, delta=1))
run("small: 2048 x u32 normal-ish n=6600", [(rng.standard_normal(6600) * 1e6).astype(np.int32) for _ in range(2048)], dict(mode=1, delta=1))
