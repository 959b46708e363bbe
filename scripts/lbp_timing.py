"""Per-stage busy cycles of enc_lookback_pipe_kernel (needs a -DPCO_LBP_TIMING build: scripts/build_variant.sh lbptiming -DPCO_LBP_TIMING,
then PCO_GFX_LIB=ab/libpco_gfx_lbptiming.so python scripts/lbp_timing.py)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import gpu_util as U
from pcodec_amd import _lib as G
L = G.lib()
names = ["H0 (fine table)", "H1 (coarse table)", "C  (candidates)", "D  (decisions)", "A  (apply)"]
def run(tag, arrays, kw):
    z = (C.c_ulonglong * 16)()
    U.gpu_batched(arrays[:8], G.make_config(**kw))
    L.pco_gfx_debug_lbp_timing(z, 1)
    L.pco_gfx_profile_begin()
    U.gpu_batched(arrays, G.make_config(**kw))
    names_b = C.create_string_buffer(1 << 16); ms = (C.c_float * 4096)()
    nk = L.pco_gfx_profile_end(names_b, len(names_b), ms, 4096)
    raw = names_b.raw; pos = 0; kt = {}
    for i in range(nk):
        e = raw.index(b"\0", pos); kt[raw[pos:e].decode()] = kt.get(raw[pos:e].decode(), 0) + ms[i]; pos = e + 1
    L.pco_gfx_debug_lbp_timing(z, 0)
    v = list(z); tiles = max(v[6], 1)
    lbk = {k: round(t, 3) for k, t in kt.items() if "lookback" in k}
    print(tag, "pages", v[7], "tiles/page", round(v[6] / max(v[7], 1), 1), "rounds/tile", round(v[5] / tiles, 2), lbk)
    for k in range(5): print(f"   {names[k]:24s} {v[k] / tiles:10.0f} busy cycles per step")
    print(f"   D: load {v[8] / tiles:.0f}, slots {v[9] / tiles:.0f}, rounds {v[10] / tiles:.0f}; barrier wait D {v[11] / tiles:.0f}, C {v[12] / tiles:.0f}; fast tiles {v[13] / tiles:.3f}")
    print(f"   hash pre-pass (per 15-tile step): worker 0 busy {v[14] / max(v[0], 1):.0f}, sequencer busy {v[15] / max(v[0], 1):.0f} cycles")
rng = np.random.default_rng(1)
run("i64 seasonal x64", [U.synth("c4", seed=s) for s in range(64)], dict(mode=1, delta=3))
run("i64 seasonal x1024", [U.synth("c4", seed=s) for s in range(1024)], dict(mode=1, delta=3))
run("u32 random x256", [rng.integers(0, 1 << 32, 1 << 18, dtype=np.uint64).astype(np.uint32) for _ in range(256)], dict(mode=1, delta=3))
run("trial-sized: 2048 x u64 random ints 1000..10000 n=6563", [rng.integers(1000, 10000, 6563).astype(np.uint64) for _ in range(2048)], dict(mode=1, delta=3))
run("trial-sized: 2048 x u64 noisy ramp n=6563", [U.synth("c2", seed=s)[:6563] for s in range(2048)], dict(mode=1, delta=3))
