"""Diagnostics for the adversarial histogram fixtures: bins of the GPU's default and strict modes against the oracle's two rules."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as U, oracle_lib as O
from pcodec_amd import _lib as G
fx = np.load(os.path.join(ROOT, "tests", "golden", "hist_fallback.npz"))
kw = dict(mode=1, delta=1)
for key in ("n5000", "n262144"):
    x = fx[key]
    lit = O.simple_compress(x, O.make_config(**kw))
    O.set_hist_rule(1); ms = O.simple_compress(x, O.make_config(**kw)); O.set_hist_rule(0)
    for rep in range(2):
        d = U.gpu_simple_compress(x, G.make_config(**kw)); s = U.gpu_simple_compress(x, G.make_config(strict_histogram=True, **kw))
        il, bl = O.inspect_first_chunk(lit); im, bm = O.inspect_first_chunk(ms); idf, bd = O.inspect_first_chunk(d); ist, bs = O.inspect_first_chunk(s)
        print(key, rep, "n_bins lit/ms/default/strict", il.n_bins[1], im.n_bins[1], idf.n_bins[1], ist.n_bins[1], "default==ms", d == ms, "default==lit", d == lit, "strict==lit", s == lit, "strict==ms", s == ms, flush=True)
        if d != ms:
            a, b = np.asarray(bd[1]), np.asarray(bm[1])
            k = min(len(a), len(b)); dif = [i for i in range(k) if not np.array_equal(a[i], b[i])][:5]
            print("  first differing bins (default vs multiset):", dif, [ (a[i].tolist(), b[i].tolist()) for i in dif[:3]])
