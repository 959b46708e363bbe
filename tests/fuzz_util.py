"""Randomised parity sweep shared by tests/test_gpu_parity.py and scripts/gpu_fuzz.py: random dtype / size / distribution /
ChunkConfig; the GPU's bytes must equal the oracle's (unless the reference's order-dependent heapsort fallback ran) and the
GPU must decode the oracle's bytes to the input."""
import numpy as np

import oracle_lib as O
import gpu_util as U
from pcodec_amd import _lib as G

INT = [np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.uint64, np.int64]
FLT = [np.float16, np.float32, np.float64]


def gen(rng, dt, n):
    kind = rng.integers(0, 9)
    if np.dtype(dt).kind == "f":
        if kind == 0: x = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 6)
        elif kind == 1: x = rng.integers(0, 5000, n) / 100.0
        elif kind == 2: x = np.round(rng.standard_normal(n) * 100) * 0.25
        elif kind == 3: x = np.cumsum(rng.standard_normal(n))
        elif kind == 4: x = rng.choice([0.0, -0.0, 1.5, np.inf, -np.inf, np.nan, 1e30, 1e-30], n)
        elif kind == 5: x = rng.integers(-1000, 1000, n).astype(np.float64)
        elif kind == 6: x = np.full(n, 3.25)
        elif kind == 7: x = rng.integers(0, 100, n) * 0.1 + rng.standard_normal(n) * 1e-9
        else: x = np.frombuffer(rng.bytes(n * 8), np.uint64).astype(np.float64)
        return x.astype(dt)
    ii = np.iinfo(dt); lo, hi = int(ii.min), int(ii.max)
    cap = min(hi, (1 << 62) - 1)   # modulus that fits every intermediate int64
    if kind == 0: x = rng.integers(lo, hi, n, dtype=np.int64 if hi < 2 ** 63 else np.uint64, endpoint=True) if dt != np.uint64 else rng.integers(0, hi, n, dtype=np.uint64, endpoint=True)
    elif kind == 1: x = rng.integers(0, min(300, cap), n)
    elif kind == 2: x = np.cumsum(rng.integers(-3, 9, n)) % (min(hi, 1 << 40) + 1)
    elif kind == 3: x = (rng.integers(0, 50, n) * int(rng.integers(2, 100))) % (cap + 1)
    elif kind == 4: x = np.where(rng.random(n) < 0.8, 7, rng.integers(0, min(cap, 1 << 40), n))
    elif kind == 5: x = np.arange(n) % (cap + 1)
    elif kind == 6: x = rng.geometric(0.02, n) % (cap + 1)
    elif kind == 7:
        base = rng.integers(0, min(cap, 1 << 30), 37); x = base[np.arange(n) % 37] + rng.integers(0, 3, n)
        x = x % (cap + 1)
    else: x = np.full(n, int(rng.integers(0, cap)))
    return np.asarray(x).astype(dt)



def draw_config(rng, dt, n, max_level=8):
    isf = np.dtype(dt).kind == "f"
    kw = dict(level=int(rng.integers(0, max_level + 1)))
    m = rng.integers(0, 5)
    if m == 0: kw["mode"] = 0
    elif m == 1: kw["mode"] = 1
    elif m == 2 and not isf: kw.update(mode=4, mode_u64=int(rng.integers(1, 1000)))
    elif m == 3 and isf: kw.update(mode=2, mode_f64=float(rng.choice([0.01, 0.1, 0.25, 1.0, 3.0])))
    elif m == 4 and isf: kw.update(mode=3, mode_u64=int(rng.integers(1, 20 if np.dtype(dt).itemsize >= 4 else 10)))
    else: kw["mode"] = 1
    d = rng.integers(0, 5)
    if d == 0: kw["delta"] = 0
    elif d == 1: kw["delta"] = 1
    elif d in (2, 3): kw.update(delta=2, delta_order=int(rng.integers(1, 8)))
    else: kw["delta"] = 3
    return kw


def page_sizes(n, max_page_n):
    """PagingSpec::EqualPagesUpTo (chunk_config.rs:134-182): what standalone::simple_compress cuts the input into."""
    if n == 0: return []
    k = -(-n // max_page_n); low, r = divmod(n, k)
    return [low + 1] * r + [low] * (k - r)


def hist_fallback_ran(nums, kw):
    """Did the reference's order-dependent heapsort fallback (histograms.rs:248-258) run for any chunk of this input?"""
    sizes = page_sizes(nums.size, kw["max_page_n"]) if kw.get("max_page_n") else [nums.size]
    pos = 0
    for sz in sizes:
        sub_kw = {k: v for k, v in kw.items() if k != "max_page_n"}
        _, _, fb = O.chunk_plan(nums[pos: pos + sz], O.make_config(enable_8_bit=True, **sub_kw))
        if fb: return True
        pos += sz
    return False


SIZES = [1, 2, 3, 17, 255, 256, 257, 513, 1025, 1000, 4099, 20000, 70000, 1 << 18, (1 << 18) + 1]
SIZE_P = [.04, .03, .03, .04, .04, .04, .05, .04, .04, .18, .18, .17, .08, .02, .02]


def run(n_cases, seed, only_8bit=False, max_level=8, strict=False):
    """Returns (bad, skipped, fails): `skipped` maps a reason to the number of cases that were not compared -- callers assert a
    budget on it (a regression that refuses more inputs must not pass as green).  strict: the GPU runs with PCO_GFX_CFG_STRICT_HISTOGRAM,
    and then its bytes are compared with the oracle's whatever branch the reference's histogram took (no fallback skip)."""
    rng = np.random.default_rng(seed)
    bad = []; skipped = {}; fails = {}

    def skip(why): skipped[why] = skipped.get(why, 0) + 1

    for case in range(n_cases):
        dt = (INT + FLT)[rng.integers(0, 11)] if not only_8bit else INT[rng.integers(0, 2)]
        n = int(rng.choice(SIZES, p=SIZE_P))
        nums = gen(rng, dt, n)
        kw = draw_config(rng, dt, n, max_level)
        if rng.random() < 0.15: kw["max_page_n"] = int(rng.integers(1, max(n, 2))) if n < 100000 else int(rng.integers(1 << 16, n))
        try:
            ocfg = O.make_config(enable_8_bit=True, **kw)
            want = O.simple_compress(nums, ocfg)
            fb = hist_fallback_ran(nums, kw)
        except O.OracleError as e:
            try:
                U.gpu_simple_compress(nums, G.make_config(enable_8_bit=True, **kw)); bad.append(("gpu accepted what the oracle refused", case, np.dtype(dt).name, n, kw))
            except G.PcoGfxError:
                pass
            skip("oracle refused: " + str(e)[:60]); continue
        try:
            got = U.gpu_simple_compress(nums, G.make_config(enable_8_bit=True, strict_histogram=strict, **kw))
        except G.PcoGfxError as e:
            if e.status == G.ST_UNSUPPORTED: skip("gpu unsupported: " + str(e)[-60:]); continue
            bad.append(("gpu error", case, np.dtype(dt).name, n, kw, str(e))); continue
        if got != want:
            if not fb or strict:
                bad.append(("bytes", case, np.dtype(dt).name, n, kw))
                fails[f"case{case}_nums"] = nums; fails[f"case{case}_got"] = np.frombuffer(got, np.uint8); fails[f"case{case}_kw"] = np.array(repr(kw))
                continue
            # the reference's order-dependent histogram branch ran: the bytes may differ, but they must
            # still be a valid encoding of the input for both decoders
            skip("bytes not compared: reference heapsort-fallback histogram")
            try:
                if not U.bits_equal(U.gpu_simple_decompress(got, nums.dtype, n), nums) or not U.bits_equal(O.simple_decompress(got, nums.dtype, cap=n + 8), nums):
                    bad.append(("divergent bytes do not decode to the input", case, np.dtype(dt).name, n, kw))
            except (G.PcoGfxError, O.OracleError) as e:
                bad.append(("divergent bytes fail to decode", case, np.dtype(dt).name, n, kw, str(e)))
        try:
            back = U.gpu_simple_decompress(want, nums.dtype, n)
            if not U.bits_equal(back, nums): bad.append(("decode", case, np.dtype(dt).name, n, kw))
        except G.PcoGfxError as e:
            if e.status != G.ST_UNSUPPORTED: bad.append(("decode error", case, np.dtype(dt).name, n, kw, str(e)))
            else: skip("gpu decode unsupported: " + str(e)[-60:])

    return bad, skipped, fails


def run_batched(n_calls, seed, max_level=8, strict=False):
    """Many chunks of mixed dtype / size / distribution in ONE pco_gfx_compress_chunks call (the path the benchmark and a
    row-group writer use): every chunk's bytes against the oracle, every chunk decoded back by the batched decoder."""
    rng = np.random.default_rng(seed)
    bad = []; skipped = {}
    for call in range(n_calls):
        k = int(rng.integers(2, 40))
        dts = [(INT[2:] + FLT[1:])[rng.integers(0, 8)] for _ in range(k)]
        ns = [int(rng.choice(SIZES, p=SIZE_P)) for _ in range(k)]
        arrays = [gen(rng, dt, n) for dt, n in zip(dts, ns)]
        kw = draw_config(rng, np.uint32, 0, max_level)
        if kw.get("mode") not in (0, 1): kw["mode"] = int(rng.integers(0, 2))   # a mode every dtype accepts
        try:
            chunks, back = U.gpu_batched(arrays, G.make_config(strict_histogram=strict, **kw))
        except G.PcoGfxError as e:
            if e.status == G.ST_UNSUPPORTED: skipped["gpu unsupported: " + str(e)[-60:]] = skipped.get("gpu unsupported: " + str(e)[-60:], 0) + 1; continue
            bad.append(("gpu error", call, kw, str(e))); continue
        for i, a in enumerate(arrays):
            # (a batched task is ONE chunk whatever its size; the oracle's file writer needs max_page_n >= n to keep it in one)
            want = O.simple_compress(a, O.make_config(max_page_n=max(a.size, 1 << 18), **kw))
            if chunks[i] != U.chunk_of_file(want, len(chunks[i])) and (strict or not hist_fallback_ran(a, kw)):
                bad.append(("bytes", call, i, a.dtype.name, a.size, kw))
            if not U.bits_equal(back[i], a): bad.append(("decode", call, i, a.dtype.name, a.size, kw))
    return bad, skipped
