"""Generate tests/golden/oracle_fixtures.npz: small seeded inputs and the .pco bytes the ORACLE produces
for them (the oracle itself is pinned against the reference's assets, see tests/test_oracle_golden.py).
The GPU tests must reproduce these bytes exactly.  Run from the repo root: python tests/golden/make_fixtures.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

CASES = []


def add(name, nums, **cfg):
    CASES.append((name, np.ascontiguousarray(nums), cfg))


rng = np.random.default_rng(20260925)
n = 3000
add("u64_ramp_delta1", np.uint64(1 << 40) + np.uint64(1000) * np.arange(n, dtype=np.uint64) + rng.integers(0, 512, n).astype(np.uint64), mode=1, delta=2, delta_order=1)
add("u32_uniform_classic", rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32), mode=1, delta=1)
add("f64_decimal_floatmult", rng.integers(1000, 10000, n) / 100.0, mode=2, mode_f64=0.01, delta=1)
add("f32_decimal_floatmult_delta1", (rng.integers(1000, 10000, n) / 100.0).astype(np.float32), mode=2, mode_f64=0.01, delta=2, delta_order=1)
add("i64_seasonal_lookback", (rng.integers(-(1 << 40), 1 << 40, 365)[np.arange(n) % 365] + rng.integers(-3, 4, n)).astype(np.int64), mode=1, delta=3)
add("i32_delta2", np.cumsum(np.cumsum(rng.integers(-3, 4, n))).astype(np.int32), mode=1, delta=2, delta_order=2)
add("i64_intmult8", (rng.integers(-1000, 1000, n) * 8 - 1).astype(np.int64), mode=4, mode_u64=8, delta=1)
add("f32_quant12", (rng.standard_normal(n) * 100).astype(np.float16).astype(np.float32), mode=3, mode_u64=13, delta=1)
add("u16_classic_delta1", (np.arange(n) * 3 + rng.integers(0, 5, n)).astype(np.uint16), mode=1, delta=2, delta_order=1)
add("u64_multichunk", np.uint64(7) * np.arange(2500, dtype=np.uint64) ** 2, mode=1, delta=2, delta_order=2, max_page_n=1000)
add("f64_sparse_zero", np.where(rng.random(n) < 0.9, 0.0, rng.standard_normal(n)), mode=1, delta=1)
add("u32_tiny", np.array([5, 5, 6], np.uint32), mode=1, delta=2, delta_order=1)
add("u64_incompressible_delta1_fallback", rng.integers(0, 1 << 63, 1200, dtype=np.uint64), mode=1, delta=2, delta_order=1)
add("i64_auto_auto", np.cumsum(rng.integers(-5, 50, n)).astype(np.int64) * 4)
add("f64_auto_auto", rng.integers(1000, 10000, n) / 100.0)

out = {}
for name, nums, cfg in CASES:
    enc = O.simple_compress(nums, O.make_config(**cfg))
    out[name + "__nums"] = nums
    out[name + "__pco"] = np.frombuffer(enc, np.uint8)
    out[name + "__cfg"] = np.array([cfg.get("mode", 0), cfg.get("mode_u64", 0), cfg.get("delta", 0), cfg.get("delta_order", 0), cfg.get("max_page_n", 0)], np.int64)
    out[name + "__f64"] = np.array([cfg.get("mode_f64", 0.0)])
np.savez_compressed(os.path.join(HERE, "oracle_fixtures.npz"), **out)
print("wrote", len(CASES), "cases", sum(v.nbytes for v in out.values()), "bytes")
