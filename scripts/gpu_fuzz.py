"""Randomised parity sweep on the GPU box (tests/fuzz_util.py).  usage: python scripts/gpu_fuzz.py [n_cases] [seed]; FUZZ_8BIT=1
restricts it to u8 / i8.  Failing inputs are saved to gpurun_out/fuzz_fail_<seed>.npz."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import fuzz_util

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad, skipped, fails = fuzz_util.run(n_cases, seed, only_8bit=os.environ.get("FUZZ_8BIT") is not None, max_level=12)
print(f"cases {n_cases} skipped {sum(skipped.values())} {skipped} bad {len(bad)}")
if fails:
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out")
    os.makedirs(out, exist_ok=True)
    np.savez(os.path.join(out, f"fuzz_fail_{seed}.npz"), **fails)
for b in bad[:25]: print(b)
