"""Debug helper: decode a few streams through the fast and the legacy decoder and report first mismatches."""
import os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle_lib as O
import gpu_util as U
from test_oracle_golden import EXPECTED, asset

def check(name, data, exp):
    got = U.gpu_simple_decompress(data, exp.dtype, max(exp.size, 1))
    ok = U.bits_equal(got, exp)
    if not ok:
        if got.shape != exp.shape:
            print(name, "shape", got.shape, exp.shape)
        else:
            bad = np.nonzero(got.view(np.uint8).reshape(exp.size, -1) != exp.view(np.uint8).reshape(exp.size, -1))[0]
            print(name, "MISMATCH n=", exp.size, "first bad idx", bad[:8], "count", len(np.unique(bad)), "got", got[bad[:4]], "exp", exp[bad[:4]])
    else:
        print(name, "ok")

for name, exp in sorted(EXPECTED.items()):
    if exp.dtype.itemsize == 1: continue
    try: check(name, asset(name), exp)
    except Exception as e: print(name, "EXC", e)
rng = np.random.default_rng(0)
for n in [1, 5, 255, 256, 257, 600, 2000, 4096, 70000]:
    for dt in [np.uint32, np.uint64, np.int16]:
        for (dk, do) in [(1, 0), (2, 1), (2, 2)]:
            nums = rng.integers(0, 1000, n).astype(dt)
            enc = O.simple_compress(nums, O.make_config(delta=dk, delta_order=do))
            info, bins = O.inspect_first_chunk(enc)
            print("   asl", list(info.ans_size_log), "nbins", list(info.n_bins), end="  ")
            try: check(f"n={n} {np.dtype(dt).name} d{dk}/{do}", enc, nums)
            except Exception as e: print(n, dt, dk, do, "EXC", e)
