import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle_lib as O
import gpu_util as U
import fuzz_util as F
from pcodec_amd import _lib as G
from fuzz_util import *

rng = np.random.default_rng(77)
for call in range(12):
    k = int(rng.integers(2, 40))
    dts = [(INT[2:] + FLT[1:])[rng.integers(0, 8)] for _ in range(k)]
    ns = [int(rng.choice(SIZES, p=SIZE_P)) for _ in range(k)]
    arrays = [gen(rng, dt, n) for dt, n in zip(dts, ns)]
    kw = draw_config(rng, np.uint32, 0, 8)
    if kw.get("mode") not in (0, 1): kw["mode"] = int(rng.integers(0, 2))
    if call not in (3, 5, 11): continue
    def check(tag, arrs):
        chunks, back = U.gpu_batched(arrs, G.make_config(**kw))
        res = []
        for i, a in enumerate(arrs):
            want = O.simple_compress(a, O.make_config(**kw))
            w = U.chunk_of_file(want, len(chunks[i]))
            if chunks[i] != w:
                aa = np.frombuffer(chunks[i], np.uint8); bb = np.frombuffer(w, np.uint8); m = min(len(aa), len(bb)); d = np.nonzero(aa[:m] != bb[:m])[0]
                info, bins = O.inspect_first_chunk(want)
                res.append((i, a.dtype.name, a.size, len(chunks[i]), len(want) , int(d[0]) if len(d) else -1, len(d), int(info.meta_end_byte), list(info.n_bins), info.delta_kind, info.delta_order, info.mode_kind))
        print(tag, kw, "bad:", res)
    check(f"call {call} full ({k} chunks: {[ (a.dtype.name, a.size) for a in arrays]})", arrays)
    bad_i = {3: 33, 5: 22, 11: 19}[call]
    check(f"call {call} only chunk {bad_i}", [arrays[bad_i]])
    check(f"call {call} chunk {bad_i} first", [arrays[bad_i]] + arrays[:bad_i])
    check(f"call {call} big chunks only", [a for a in arrays if a.size > 100000])
    check(f"call {call} same width only", [a for a in arrays if a.dtype.itemsize == arrays[bad_i].dtype.itemsize])
    got = U.gpu_simple_compress(arrays[bad_i], G.make_config(**kw)); print("  single-call path equal:", got == O.simple_compress(arrays[bad_i], O.make_config(**kw)))
