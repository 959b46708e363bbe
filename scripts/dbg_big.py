import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle_lib as O, gpu_util as U
from pcodec_amd import _lib as G
rng = np.random.default_rng(5)
n = 20000
datasets = {
    "narrow": rng.integers(0, 300, n).astype(np.uint64),
    "wide": rng.integers(0, 1 << 62, n, dtype=np.uint64),
    "ties": np.where(rng.random(n) < 0.6, 7, rng.integers(0, 1 << 40, n)).astype(np.uint64),
    "normal_f32": rng.standard_normal(n).astype(np.float32),
}
for level in (9, 10, 12):
    for name, nums in datasets.items():
        kw = dict(level=level, mode=1, delta=1)
        want = O.simple_compress(nums, O.make_config(**kw))
        try: got = U.gpu_simple_compress(nums, G.make_config(**kw))
        except G.PcoGfxError as e: print(level, name, "ERR", e); continue
        iw, bw = O.inspect_first_chunk(want)
        try: ig, bg = O.inspect_first_chunk(got)
        except Exception as e: print(level, name, "GPU bytes unparsable", e, len(got), len(want)); continue
        same = got == want
        print(level, name, "same" if same else "DIFF", "len", len(got), len(want), "nbins", list(ig.n_bins), list(iw.n_bins), "asl", list(ig.ans_size_log), list(iw.ans_size_log))
        if not same and ig.n_bins[1] == iw.n_bins[1]:
            d = np.nonzero((bg[1] != bw[1]).any(axis=1))[0]
            print("    first differing bins", d[:5], bg[1][d[:3]].tolist(), bw[1][d[:3]].tolist())
        elif not same:
            print("    gpu bins head", bg[1][:4].tolist(), "oracle", bw[1][:4].tolist())
