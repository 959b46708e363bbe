"""Hunting a timing-dependent fault of enc_walkp_kernel: one array, paged, encoded `reps` times through the standalone entry point; reports which pages differ from the oracle.
usage: dbg_wp_pages.py <dtype> <n> <max_page_n> <value span> [reps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import oracle_lib as O, gpu_util as U
from pcodec_amd import _lib as G
dt, n, mp, span = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]); reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
rng = np.random.default_rng(7)
nums = rng.integers(0, span, n).astype(dt) if span > 0 else np.resize(np.load(os.path.join(os.path.dirname(__file__), 'data', 'case91_u8.npy')), n).astype(dt)
kw = dict(level=7, mode=1, delta=1, max_page_n=mp)
want = np.frombuffer(O.simple_compress(nums, O.make_config(enable_8_bit=True, **kw)), np.uint8)
n_bad = 0; where = {}
for r in range(reps):
    got = np.frombuffer(U.gpu_simple_compress(nums, G.make_config(enable_8_bit=True, **kw)), np.uint8)
    if len(got) != len(want) or (got != want).any():
        n_bad += 1
        if len(got) == len(want):
            idx = np.nonzero(got != want)[0]
            pages = sorted(set(int(i) * ((n + mp - 1) // mp) // len(want) for i in idx)); where[("pages~", tuple(pages))] = where.get(("pages~", tuple(pages)), 0) + 1
            key = (int(idx[0]) * 1000 // len(want), int(idx[-1]) * 1000 // len(want)); where[key] = where.get(key, 0) + 1
        else: where["len"] = where.get("len", 0) + 1
        if len(got) == len(want) and n_bad <= 3:
            xb = np.unpackbits(got ^ want, bitorder='little'); pos = np.nonzero(xb)[0]
            print('   differing bits', len(pos), 'from bit', pos[0], 'to', pos[-1], '(span', pos[-1] - pos[0], ') gaps histogram:', np.bincount(np.minimum(np.diff(pos), 40))[:41].tolist())
            print('   first 60 rel positions', (pos[:60] - pos[0]).tolist())
if hasattr(G.lib(), 'pco_gfx_debug_wp_err'):
    import ctypes as C
    e = (C.c_uint32 * 8)(); G.lib().pco_gfx_debug_wp_err(e); print('   assertion counters (scan, fields changed, stage dirty):', list(e)[:4])
print(dt, n, "pages of", mp, "span", span, "->", len(want), "bytes;", n_bad, "of", reps, "runs differ; (first, last) differing byte in permille of the file:", sorted(where.items(), key=lambda t: -t[1])[:6])
