"""How often does the reference's order-dependent heapsort fallback of the histogram (histograms.rs:248-258) run?

The GPU implements the quickselect path's result, a pure function of the sorted multiset; when the reference
falls back to heapsort + apply_sorted the bytes may differ on ties.  This census runs the ORACLE's literal restatement (which
reports whether the fallback ran, chunk_plan(...).hist_fallback) over the reference's own synthetic distributions
(pco_cli/generate_randoms.py, re-stated here with seeded generators at the chunk size 2^18) and over adversarial orders.

usage: python scripts/hist_fallback_census.py [seeds_per_case] [out.json]     (CPU only)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle_lib as O

N = int(os.environ.get("CENSUS_N", 1 << 18))
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
out_path = sys.argv[2] if len(sys.argv) > 2 else None


def lomax(rng, a, median, n): return rng.pareto(a, n) / (2 ** (1 / a) - 1) * median


def money(rng, n):
    dollars = np.floor(lomax(rng, 1.5, 5, n)).astype(np.int64); cents = rng.integers(0, 100, n); p = rng.uniform(size=n)
    for thr, v in ((0.9, 99), (0.75, 98), (0.6, 95), (0.45, 75), (0.4, 50), (0.25, 25), (0.15, 0)): cents[p < thr] = v
    return dollars, cents, dollars * 100 + cents


GENS = {   # name: (function(rng, n) -> float/int array, dtypes)  -- generate_randoms.py:157-380
    "geo": (lambda r, n: r.geometric(0.001, n), ["i64"]),
    "lomax05": (lambda r, n: lomax(r, 0.5, 1000, n), ["i32", "u32", "i64"]),
    "uniform": (lambda r, n: r.integers(-(2 ** 63), 2 ** 63, n, dtype=np.int64), ["i64"]),
    "constant": (lambda r, n: np.repeat(77777, n), ["i64"]),
    "sparse": (lambda r, n: r.binomial(1, 0.01, n), ["i64"]),
    "dollars": (lambda r, n: money(r, n)[0], ["i64"]),
    "cents": (lambda r, n: money(r, n)[1], ["u8", "i16", "i64"]),
    "total_cents": (lambda r, n: money(r, n)[2], ["i64"]),
    "slow_cosine": (lambda r, n: 100_000 * np.cos(np.arange(n) * 2 * np.pi / (n / 103)), ["i64", "f64"]),
    "normal": (lambda r, n: r.normal(size=n), ["f64", "f32", "f16"]),
    "log_normal": (lambda r, n: np.exp(r.normal(size=n)), ["f32"]),
    "csum": (lambda r, n: np.cumsum(np.exp(r.normal(size=n)) - np.exp(0.5)), ["f32"]),
    "near_linear": (lambda r, n: 10 ** 6 * (1640995200 + np.arange(n) + r.normal(size=n)), ["i64"]),
    "millis": (lambda r, n: 10 ** 3 * (1640995200000 + r.integers(0, 10 ** 9, n, dtype=np.int64)), ["i64"]),
    "integers": (lambda r, n: r.integers(0, 2 ** 30, n), ["i64"]),
    "quantized_normal": (lambda r, n: r.normal(size=n).astype(np.float32).astype(np.float64), ["f64"]),
    "decimal": (lambda r, n: r.integers(1000, 10000, n) / 100, ["f64", "f32"]),
    "radians": (lambda r, n: r.integers(0, 360, n) * np.pi / 180, ["f64", "f32"]),
    "interl0": (lambda r, n: (r.integers(0, 10 ** 6, 10)[None, :] + r.normal(scale=22, size=[n // 10, 10])).reshape(-1), ["i64", "f32"]),
    "interl1": (lambda r, n: np.cumsum(r.integers(-10, 10, [n // 10, 10]), axis=0).reshape(-1) + 10 ** 6, ["i64"]),
    "dist_shift": (lambda r, n: 0.5 + np.exp(np.linspace(-4, 4, n)) * r.normal(size=n), ["f32"]),
    "ids": (lambda r, n: r.choice(r.integers(0, 2 ** 63, 1000, dtype=np.int64), n, p=(lambda w: w / w.sum())(np.exp(r.uniform(0, 5, 1000)))), ["i64"]),
    # adversarial orders for a quickselect: sorted, reversed, sawtooth, organ pipe, heavy ties
    "sorted_uniform": (lambda r, n: np.sort(r.integers(0, 2 ** 40, n)), ["i64"]),
    "reversed_uniform": (lambda r, n: np.sort(r.integers(0, 2 ** 40, n))[::-1].copy(), ["i64"]),
    "sawtooth": (lambda r, n: np.arange(n) % 1000 + r.integers(0, 3, n), ["i64"]),
    "organ_pipe": (lambda r, n: np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]]) * 1000003, ["i64"]),
    "ties_60pct": (lambda r, n: np.where(r.random(n) < 0.6, 7, r.integers(0, 2 ** 40, n)), ["i64"]),
    "two_values": (lambda r, n: r.integers(0, 2, n) * (2 ** 40), ["i64"]),
    "ramp_noise": (lambda r, n: 2 ** 40 + 1000 * np.arange(n) + r.integers(0, 512, n), ["u64"]),
}
NP = {"i64": np.int64, "u64": np.uint64, "i32": np.int32, "u32": np.uint32, "i16": np.int16, "u8": np.uint8, "f64": np.float64, "f32": np.float32, "f16": np.float16}
CFGS = {"auto": dict(), "classic_nodelta": dict(mode=1, delta=1), "classic_delta1": dict(mode=1, delta=2, delta_order=1), "classic_lookback": dict(mode=1, delta=3)}

res = {}; t0 = time.time(); total = fired = 0
for name, (fn, dts) in GENS.items():
    for dt in dts:
        for cname, kw in CFGS.items():
            if cname == "classic_lookback" and name not in ("sawtooth", "interl0", "interl1", "ids", "cents"): continue
            k = 0; err = 0
            for s in range(seeds):
                rng = np.random.default_rng(1000 * s + 17)
                x = np.asarray(fn(rng, N))
                x = x.astype(NP[dt]) if x.dtype.kind != "f" or NP[dt](0).dtype.kind == "f" else np.clip(x, np.iinfo(NP[dt]).min, np.iinfo(NP[dt]).max).astype(NP[dt])
                try:
                    _, _, fb = O.chunk_plan(x, O.make_config(enable_8_bit=True, **kw))
                except O.OracleError:
                    err += 1; continue
                k += int(fb)
            res[f"{name}/{dt}/{cname}"] = {"fallback_ran": k, "chunks": seeds - err, "refused": err}
            total += seeds - err; fired += k
            if k: print(f"{name}/{dt}/{cname}: heapsort fallback ran in {k} of {seeds - err} chunks", flush=True)
print(f"{fired} of {total} chunks took the heapsort fallback ({time.time() - t0:.0f}s, {seeds} seeds per case, n = {N})")
if out_path:
    json.dump({"n": N, "seeds_per_case": seeds, "chunks": total, "fallback_ran": fired, "cases": res}, open(out_path, "w"), indent=1)
