"""enc_lookback_kernel time against the number of pages in one call (how many pages' tables are live at once)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import gpu_util as U
from pcodec_amd import _lib as G
L = G.lib()
def run(tag, arrays, kw):
    U.gpu_batched(arrays[:8], G.make_config(**kw))
    L.pco_gfx_profile_begin()
    U.gpu_batched(arrays, G.make_config(**kw))
    names_b = C.create_string_buffer(1 << 16); ms = (C.c_float * 4096)()
    nk = L.pco_gfx_profile_end(names_b, len(names_b), ms, 4096)
    raw = names_b.raw; pos = 0; kt = {}
    for i in range(nk):
        e = raw.index(b"\0", pos); kt[raw[pos:e].decode()] = kt.get(raw[pos:e].decode(), 0) + ms[i]; pos = e + 1
    t = sum(v for k, v in kt.items() if "lookback" in k)
    n = sum(a.size for a in arrays)
    print(f"{tag:40s} pages {len(arrays):5d}  lookback kernel {t:8.3f} ms  {n / t / 1e6:7.2f} G elem/s")
base = [U.synth("c4", seed=s) for s in range(64)]
for k in (64, 128, 256, 384, 512, 768, 1024, 2048):
    run("i64 seasonal 2^18", [base[i % 64] for i in range(k)], dict(mode=1, delta=3))
