import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle_lib as O
import gpu_util as U
import fuzz_util as F
from pcodec_amd import _lib as G

def diff(tag, got, want):
    if got == want: return True
    a = np.frombuffer(got, np.uint8); b = np.frombuffer(want, np.uint8); m = min(len(a), len(b)); d = np.nonzero(a[:m] != b[:m])[0]
    info, bins = O.inspect_first_chunk(want)
    print(tag, "DIFF len", len(got), len(want), "first", int(d[0]) if len(d) else m, "ndiff", len(d), "meta_end", info.meta_end_byte, "asl", list(info.ans_size_log), "nbins", list(info.n_bins), "delta", info.delta_kind, info.delta_order)
    return False

rng = np.random.default_rng(5)
for dt in (np.uint16, np.int16, np.uint32, np.uint8):
    for n in (262145, 262144, 65537, 70001):
        for kind in range(9):
            # reproduce gen's kinds deterministically
            class R:  # force a kind
                def __init__(s, r, k): s.r, s.k, s.first = r, k, True
                def integers(s, *a, **kw):
                    if s.first: s.first = False; return s.k
                    return s.r.integers(*a, **kw)
                def __getattr__(s, name): return getattr(s.r, name)
            x = F.gen(R(rng, kind), dt, n)
            for kw in (dict(mode=1, delta=2, delta_order=1), dict(mode=1, delta=2, delta_order=5), dict(mode=1, delta=1), dict(level=2, delta=2, delta_order=6)):
                want = O.simple_compress(x, O.make_config(**kw))
                try:
                    got = U.gpu_simple_compress(x, G.make_config(enable_8_bit=True, **kw))
                except G.PcoGfxError as e:
                    print(np.dtype(dt).name, n, kind, kw, "ERR", e); continue
                ok = diff(f"{np.dtype(dt).name} n={n} kind={kind} {kw}", got, want)
                if not ok:
                    _, _, fb = O.chunk_plan(x, O.make_config(**kw)); print("    hist_fallback", fb)
print("done")
