"""Replays one case of tests/fuzz_util.run(seed, only_8bit=True) several times: dbg_fuzz8.py <seed> <case> [reps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import fuzz_util as F, oracle_lib as O, gpu_util as U
from pcodec_amd import _lib as G
seed, want_case = int(sys.argv[1]), int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rng = np.random.default_rng(seed)
for case in range(want_case + 1):
    dt = F.INT[rng.integers(0, 2)]
    n = int(rng.choice(F.SIZES, p=F.SIZE_P))
    nums = F.gen(rng, dt, n)
    kw = F.draw_config(rng, dt, n, 8)
    if rng.random() < 0.15: kw["max_page_n"] = int(rng.integers(1, max(n, 2))) if n < 100000 else int(rng.integers(1 << 16, n))
    if case != want_case: continue
    if os.environ.get('DBG_DELTA'): kw = dict(kw, delta=int(os.environ['DBG_DELTA']))
    want = O.simple_compress(nums, O.make_config(enable_8_bit=True, **kw))
    print(case, np.dtype(dt).name, n, kw, "oracle bytes", len(want), "values", nums[:12], int(nums.min()), int(nums.max()))
    for r in range(reps):
        got = U.gpu_simple_compress(nums, G.make_config(enable_8_bit=True, **kw))
        diffs = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
        print("  rep", r, "equal" if got == want else f"DIFFER: {len(diffs)} bytes, first {diffs[:6]}, last {diffs[-3:]} (lengths {len(got)} / {len(want)})")
        if got != want:
            try:
                back = O.simple_decompress(got, nums.dtype, cap=n + 8)
                bad = np.nonzero(back[:n] != nums)[0]
                print("    decoded differs at", len(bad), "numbers:", bad[:8], "...", bad[-3:], "got", back[bad[:8]], "want", nums[bad[:8]])
            except Exception as e:
                print("    decode of the GPU bytes fails:", e)
            g = np.frombuffer(got, np.uint8); w = np.frombuffer(want, np.uint8); m = min(len(g), len(w))
            x = g[:m] ^ w[:m]; idx = np.nonzero(x)[0]
            extra = int(np.sum(((g[:m] & ~w[:m]) != 0))); missing = int(np.sum(((w[:m] & ~g[:m]) != 0)))
            print("    bytes with extra one-bits", extra, "with missing one-bits", missing, "span", idx[0], idx[-1], "runs:", [(int(a[0]), int(a[-1])) for a in np.split(idx, np.nonzero(np.diff(idx) > 16)[0] + 1)][:6])
