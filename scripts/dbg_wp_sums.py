"""-DPCO_WP_DEBUGSUM build: replays fuzz case (seed 2025, case 91 of the 8-bit sweep) and compares the walker's and the packing waves' checksums per item and step."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import fuzz_util as F, oracle_lib as O, gpu_util as U
from pcodec_amd import _lib as G
seed, want_case, reps = 2025, 91, int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(seed)
for case in range(want_case + 1):
    dt = F.INT[rng.integers(0, 2)]; n = int(rng.choice(F.SIZES, p=F.SIZE_P)); nums = F.gen(rng, dt, n); kw = F.draw_config(rng, dt, n, 8)
    if rng.random() < 0.15: kw["max_page_n"] = int(rng.integers(1, max(n, 2))) if n < 100000 else int(rng.integers(1 << 16, n))
if len(sys.argv) > 2: kw = dict(kw); kw["delta"] = int(sys.argv[2])   # (1: no Auto-delta trial encodes, one run_encode per call)
want = O.simple_compress(nums, O.make_config(enable_8_bit=True, **kw))
L = G.lib(); good = None
for r in range(reps):
    got = U.gpu_simple_compress(nums, G.make_config(enable_8_bit=True, **kw))
    buf = (C.c_uint32 * (16 * 1100 * 4))(); L.pco_gfx_debug_wp_sums(buf)
    a = np.frombuffer(buf, np.uint32).reshape(16, 1100, 4)[:11, :28].copy()
    ok = got == want
    mism_f = np.argwhere(a[:, :, 0] != a[:, :, 2]); mism_s = np.argwhere(a[:, :, 1] != a[:, :, 3])
    print("rep", r, "bytes equal" if ok else "BYTES DIFFER", "| field sums walker != packer at (item, step):", mism_f.tolist()[:6], "| symbol sums:", mism_s.tolist()[:6])
    if ok and good is None: good = a
    if good is not None and not ok:
        for nm, col in (("walker fields", 0), ("walker symbols", 1), ("packer fields", 2), ("packer symbols", 3)):
            d = np.argwhere(a[:, :, col] != good[:, :, col])
            if len(d): print("    vs a good run,", nm, "differ at", d.tolist()[:6])
