"""Debug matrix for the multi-kernel encoder on two-variable chunks whose variables own different batch counts
(run with PCO_GFX_LIB pointing at a -DPCO_LOOKBACK_SEC_FAST build to reproduce the round-1 sweep failure)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle_lib as O
import gpu_util as U
from pcodec_amd import _lib as G

HERE = os.path.dirname(os.path.abspath(__file__))


def one(tag, nums, kw):
    ocfg = O.make_config(enable_8_bit=True, **kw)
    try:
        want = O.simple_compress(nums, ocfg)
    except O.OracleError as e:
        print(tag, "oracle refused", e); return
    try:
        got = U.gpu_simple_compress(nums, G.make_config(enable_8_bit=True, **kw))
    except G.PcoGfxError as e:
        print(tag, "GPU ERROR", e); return
    if got == want:
        print(tag, "ok", len(got)); return
    info, bins = O.inspect_first_chunk(want)
    a = np.frombuffer(got, np.uint8); b = np.frombuffer(want, np.uint8)
    m = min(len(a), len(b)); d = np.nonzero(a[:m] != b[:m])[0]
    first = int(d[0]) if len(d) else m
    try:
        back = O.simple_decompress(got, nums.dtype, cap=nums.size + 8)
        if back.shape == nums.shape:
            wrong = np.nonzero(back.view(np.uint8).reshape(nums.size, -1) != nums.view(np.uint8).reshape(nums.size, -1))[0]
            dec = f"decodes with {len(np.unique(wrong))} wrong numbers, first at {np.unique(wrong)[:6]}"
        else:
            dec = f"decodes to shape {back.shape}"
    except Exception as e:
        dec = f"does not decode: {e}"
    print(tag, "DIFF len got/want", len(got), len(want), "first diff byte", first, "n diff", len(d), "meta_end", info.meta_end_byte,
          "asl", list(info.ans_size_log), "nbins", list(info.n_bins), "|", dec)


def main():
    d = np.load(os.path.join(HERE, "..", "tests", "golden", "fuzz_case_2025_188.npz"))
    base = d["nums"]
    kw0 = dict(level=4, mode=4, mode_u64=134, delta=3)
    rng = np.random.default_rng(7)
    print("--- original case and size variations")
    for n in (255, 256, 257, 258, 300, 512, 513, 514, 1025):
        x = np.resize(base, n)
        one(f"u8 n={n} lookback intmult134", x, kw0)
    print("--- other dtypes, same values")
    for dt in (np.uint16, np.uint32, np.uint64, np.int32):
        for n in (257, 513):
            one(f"{np.dtype(dt).name} n={n} lookback intmult134", np.resize(base, n).astype(dt), kw0)
    print("--- non-trivial lookback variable (periodic data) with a secondary")
    per = rng.integers(0, 1 << 20, 37)
    for n in (257, 513, 1025, 300):
        x = ((per[np.arange(n) % 37] + rng.integers(0, 2, n)) * 8 + rng.integers(0, 3, n)).astype(np.uint32)
        one(f"u32 n={n} lookback(periodic) intmult8", x, dict(mode=4, mode_u64=8, delta=3))
    print("--- consecutive delta + secondary: primary stores n - order latents")
    for order in (1, 2, 7):
        for n in (256 + 1, 256 + order, 512 + 1, 512 + order, 256 + order + 1, 300):
            x = ((np.cumsum(rng.integers(-3, 9, n)) + 1000) * 8 + rng.integers(0, 3, n)).astype(np.uint32)
            one(f"u32 n={n} consec{order} intmult8", x, dict(mode=4, mode_u64=8, delta=2, delta_order=order))
            y = (rng.integers(1000, 10000, n) / 100.0)
            one(f"f64 n={n} consec{order} floatmult", y, dict(mode=2, mode_f64=0.01, delta=2, delta_order=order))
    print("--- constant primary (trivial) + noisy secondary")
    for n in (257, 513, 1000):
        x = (1000 * 8 + rng.integers(0, 8, n)).astype(np.uint32)
        for dk in (dict(delta=1), dict(delta=2, delta_order=1), dict(delta=3)):
            one(f"u32 n={n} trivial primary intmult8 {dk}", x, dict(mode=4, mode_u64=8, **dk))
    print("--- multi-page")
    x = ((np.cumsum(rng.integers(-3, 9, 5000)) + 1000) * 8 + rng.integers(0, 3, 5000)).astype(np.uint32)
    for mp in (257, 513, 1281):
        for dk in (dict(delta=2, delta_order=1), dict(delta=3)):
            one(f"u32 n=5000 max_page_n={mp} intmult8 {dk}", x, dict(mode=4, mode_u64=8, max_page_n=mp, **dk))


if __name__ == "__main__":
    main()
