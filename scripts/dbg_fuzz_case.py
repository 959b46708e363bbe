"""Replays cases of tests/fuzz_util.run(seed) and compares the GPU's bytes with the oracle's for the listed case numbers only
(the others are drawn and skipped, so that the random stream stays in step).  usage: dbg_fuzz_case.py <seed> <max_level> <case> [<case> ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import fuzz_util as F, oracle_lib as O, gpu_util as U
from pcodec_amd import _lib as G

seed, max_level = int(sys.argv[1]), int(sys.argv[2]); want_cases = set(int(a) for a in sys.argv[3:])
rng = np.random.default_rng(seed)
for case in range(max(want_cases) + 1):
    dt = (F.INT + F.FLT)[rng.integers(0, 11)]
    n = int(rng.choice(F.SIZES, p=F.SIZE_P))
    nums = F.gen(rng, dt, n)
    kw = F.draw_config(rng, dt, n, max_level)
    if rng.random() < 0.15: kw["max_page_n"] = int(rng.integers(1, max(n, 2))) if n < 100000 else int(rng.integers(1 << 16, n))
    if case not in want_cases: continue
    want = O.simple_compress(nums, O.make_config(enable_8_bit=True, **kw))
    got = U.gpu_simple_compress(nums, G.make_config(enable_8_bit=True, **kw))
    first = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), None)
    print(case, np.dtype(dt).name, n, kw, "equal" if got == want else f"DIFFER at byte {first} (lengths {len(got)} / {len(want)})", "values", nums[:12], int(nums.min()), int(nums.max()))
