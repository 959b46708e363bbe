"""-m gpu: parity of the HIP path (through the C ABI of libpco_gfx.so) against the oracle.

Bar: bit-exact.  Encode -> identical .pco bytes; decode -> identical arrays.  Every test goes through
the C ABI; the oracle is only the checker."""
import ctypes as C
import os

import numpy as np
import pytest

import gpu_util as U
import oracle_lib as O
from pcodec_amd import _lib as G

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = np.load(os.path.join(HERE, "golden", "oracle_fixtures.npz"))
FIX_NAMES = sorted({k.split("__")[0] for k in FIX.files})
# product features still on the round's to-do list (DESIGN.md "gaps"): the product must refuse them loudly
NOT_YET = {}


@pytest.fixture(scope="module")
def L():
    lib = G.lib()
    assert lib.pco_gfx_device_count() >= 1, "these tests need an MI355X; the product has no CPU path"
    return lib


def fix_cfg(name):
    c = FIX[name + "__cfg"]; f = float(FIX[name + "__f64"][0])
    kw = dict(mode=int(c[0]), mode_u64=int(c[1]), delta=int(c[2]), delta_order=int(c[3]), max_page_n=int(c[4]), mode_f64=f)
    return G.make_config(enable_8_bit=True, **kw), O.make_config(**kw)


# ------------------------------------------------------------------------------------------------ decode
def test_decode_reference_golden_assets(L):
    from test_oracle_golden import EXPECTED, asset
    from test_oracle_golden import EXPECTED_ORACLE_ONLY
    every = dict(EXPECTED); every.update(EXPECTED_ORACLE_ONLY)   # all 13 files of pco/assets: Dict mode and Conv1 delta included
    assert len(every) == 13
    for name, exp in sorted(every.items()):
        got = U.gpu_simple_decompress(asset(name), exp.dtype, max(exp.size, 1))
        assert U.bits_equal(got, exp), name
    # Dict / Conv1 streams cut short or damaged: an error, never a crash or a hang (tests/stability.rs, tests/corruption.rs)
    rng = np.random.default_rng(4)
    for name in ("v1_0_0_dict.pco", "v1_0_0_conv1.pco"):
        blob = asset(name); exp = every[name]
        for cut in list(range(0, 60)) + [len(blob) // 2, len(blob) - 2]:
            with pytest.raises(G.PcoGfxError) as ei:
                U.gpu_simple_decompress(blob[:cut], exp.dtype, exp.size)
            assert ei.value.status in (G.ST_INSUFFICIENT_DATA, G.ST_CORRUPTION), (name, cut)
        for _ in range(60):
            b = bytearray(blob); b[rng.integers(0, len(b))] ^= 1 << rng.integers(0, 8)
            try:
                U.gpu_simple_decompress(bytes(b), exp.dtype, exp.size)
            except G.PcoGfxError as e:
                assert e.status in (G.ST_CORRUPTION, G.ST_INSUFFICIENT_DATA, G.ST_INVALID_ARGUMENT, G.ST_UNSUPPORTED)


@pytest.mark.parametrize("name", FIX_NAMES)
def test_decode_committed_fixtures(L, name):
    nums = FIX[name + "__nums"]
    got = U.gpu_simple_decompress(FIX[name + "__pco"].tobytes(), nums.dtype, nums.size)
    assert U.bits_equal(got, nums)


# ------------------------------------------------------------------------------------------------ encode
@pytest.mark.parametrize("name", FIX_NAMES)
def test_encode_committed_fixtures_byte_identical(L, name):
    gcfg, _ = fix_cfg(name)
    nums = FIX[name + "__nums"]
    if name in NOT_YET:
        with pytest.raises(G.PcoGfxError) as ei:
            U.gpu_simple_compress(nums, gcfg)
        assert ei.value.status == G.ST_UNSUPPORTED, NOT_YET[name]
        return
    assert U.gpu_simple_compress(nums, gcfg) == FIX[name + "__pco"].tobytes()


@pytest.mark.parametrize("kind", ["c1", "c2", "c3", "c3d", "c4"])
def test_baseline_configs_full_size(L, kind):
    """BASELINE.json configs at n = 2^18: identical .pco bytes on encode, identical arrays on decode."""
    nums = U.synth(kind)
    gcfg, ocfg = U.cfg_pair(kind)
    want = O.simple_compress(nums, ocfg)
    got = U.gpu_simple_compress(nums, gcfg)
    assert got == want
    assert U.bits_equal(U.gpu_simple_decompress(got, nums.dtype, nums.size), nums)


def test_lookback_matrix(L):
    """delta/lookback.rs encode + decode: periodic data, hash hits beyond the LDS count cache, tiny inputs."""
    rng = np.random.default_rng(21)
    cases = []
    for dt in (np.uint32, np.int64, np.uint64):
        for n in (1, 2, 3, 17, 64, 65, 100, 1000, 9000, 40000):
            per = rng.integers(0, 1 << 30, 9 if n < 100 else 777)
            cases.append((per[np.arange(n) % len(per)] + rng.integers(0, 3, n)).astype(dt))
    far = rng.integers(0, 1 << 40, 12000); cases.append(np.tile(far, 3).astype(np.int64))   # lookbacks of 12000 > 8192
    cases.append(np.arange(3000, dtype=np.uint32) % 9)                                     # tests/recovery.rs:404-420
    for nums in cases:
        kw = dict(mode=1, delta=3)
        want = O.simple_compress(nums, O.make_config(**kw))
        got = U.gpu_simple_compress(nums, G.make_config(**kw))
        assert got == want, (nums.dtype, nums.size)
        assert U.bits_equal(U.gpu_simple_decompress(got, nums.dtype, nums.size), nums)


def test_lookback_on_duplicate_heavy_pages(L):
    """Pages of at most 8192 latents that span fewer than 4 n values (the float-mult multiples of decimal data, small counts): the
    pre-pass's screen hands them to enc_lookback_seq_kernel, which keeps the page's latents as their low 16 bits and every count as u16
    in LDS.  Every dtype, sizes around the tile and the 16-position start-up, ranges at the screen's edge, values around a 2^16 boundary
    (the low halves wrap, their difference does not)."""
    rng = np.random.default_rng(29)
    cases = []
    for dt in (np.uint16, np.int16, np.uint32, np.int32, np.int64, np.uint64):
        for n in (2, 3, 16, 17, 18, 64, 65, 66, 129, 1000, 6554, 8191, 8192):
            span = int(rng.choice([2, 7, 50, n, 2 * n, 4 * n - 1, 4 * n]))
            base = int(rng.choice([0, 65536 - span // 2, (1 << 31) - span // 2 if np.dtype(dt).itemsize >= 8 else 0]))
            if np.dtype(dt).itemsize == 2: base = int(rng.choice([0, 32768 - span // 2])) if np.dtype(dt).kind == "u" else -(span // 2)
            cases.append((base + rng.integers(0, max(span, 1), n)).astype(dt))
    for dt in (np.uint8, np.int8):   # an 8-bit page may span more than half its type: the smaller WRAPPING difference counts (lookback.rs:76-80)
        for n in (64, 256, 3000, 8192):
            cases.append(rng.integers(np.iinfo(dt).min, np.iinfo(dt).max, n, endpoint=True).astype(dt))
            cases.append(np.where(rng.random(n) < 0.8, 7, rng.integers(0, 120, n)).astype(dt))
    walk = np.cumsum(rng.integers(-1, 2, 8000)) + 70000; cases.append(walk.astype(np.int64))   # a random walk: near-duplicates next to each other
    cases.append(np.repeat(rng.integers(0, 9, 700), 9).astype(np.uint32))                       # runs
    cases.append((rng.integers(0, 40, 6554) * 3.0 + 1000.0).astype(np.float32))                 # floats through the classic mode: not screened unless close
    for nums in cases:
        kw = dict(mode=1, delta=3, enable_8_bit=True)
        want = O.simple_compress(nums, O.make_config(**kw))
        got = U.gpu_simple_compress(nums, G.make_config(**kw))
        assert got == want, (nums.dtype, nums.size, int(nums.min()), int(nums.max()))
        assert U.bits_equal(U.gpu_simple_decompress(got, nums.dtype, nums.size), nums)
    # many such pages in one call, beside pages the pipeline keeps (the redo / skip lists are per page)
    batch = [rng.integers(0, 3000, 6554).astype(np.int64) if k % 3 else (np.arange(6554) * 5 + rng.integers(0, 3, 6554)).astype(np.int64) for k in range(40)]
    chunks, back = U.gpu_batched(batch, G.make_config(mode=1, delta=3))
    for x, ch, b in zip(batch, chunks, back):
        assert ch == U.chunk_of_file(O.simple_compress(x, O.make_config(mode=1, delta=3)), len(ch)) and U.bits_equal(x, b)


def test_segmented_encode_walk_never_depends_on_luck(L):
    """enc_walkseg_kernel (encode_walkseg.hip) starts every segment of a long page from the arc of ALL tANS states and emits only once the
    arc's two ends have met; what it walked before is walked again from the true state.  Tables that forget their state at once (many
    bins), slowly (a 0.999 / 0.001 pair: hundreds of symbols) and never (equal power-of-two weights: the state is a bijection of the
    start state) must all give the reference's bytes, at page lengths with full and ragged last segments and batches."""
    rng = np.random.default_rng(31)
    cases = []
    for n in (16384, 16385, 20000, 65536 + 255, 1 << 18, (1 << 18) + 1):
        cases.append(rng.integers(0, 4, n).astype(np.uint32))                                   # four equal bins
        cases.append(rng.integers(0, 2, n).astype(np.uint32) * 1000)                            # two equal bins
        cases.append(np.where(rng.random(n) < 0.999, 5, 77).astype(np.int32))                   # one heavy bin: a rare symbol is the only thing that makes states meet
        cases.append((rng.geometric(0.3, n) % 16).astype(np.uint16))                            # a skewed handful
        cases.append(rng.integers(0, 1 << 20, n).astype(np.uint32))                             # 256 bins: meets within a few symbols
    cases.append(np.where(np.arange(1 << 18) < 200000, 5, rng.integers(0, 3, 1 << 18)).astype(np.uint32))   # a constant stretch longer than several segments
    cases.append(np.repeat(rng.integers(0, 8, 1 << 12), 64).astype(np.uint32))                  # runs
    for nums in cases:
        for kw in (dict(mode=1, delta=0), dict(mode=1, delta=2, delta_order=1)):
            want = O.simple_compress(nums, O.make_config(**kw))
            got = U.gpu_simple_compress(nums, G.make_config(**kw))
            assert got == want, (nums.dtype, nums.size, kw)
    for i in (20, 22, 29):   # the file cut into chunks of up to 50 000 numbers (simple_compress's paging)
        kw = dict(mode=1, delta=0, max_page_n=50000)
        assert U.gpu_simple_compress(cases[i], G.make_config(**kw)) == O.simple_compress(cases[i], O.make_config(**kw)), i
    # several long chunks of different tables in one call
    batch = [cases[i] for i in (0, 2, 7, 12, 24, 21)]   # (none longer than 2^18: simple_compress cuts those into two chunks)
    chunks, back = U.gpu_batched(batch, G.make_config(mode=1, delta=0))
    for x, ch, b in zip(batch, chunks, back):
        assert ch == U.chunk_of_file(O.simple_compress(x, O.make_config(mode=1, delta=0)), len(ch)) and U.bits_equal(x, b)


def test_lookback_with_two_variable_modes(L):
    """Lookback delta on the primary of int-mult / float-mult / float-quant chunks (delta_encoding.rs:303-304): the decoder keeps
    the primary latents in dst on a first pass and joins on a second."""
    rng = np.random.default_rng(23)
    for n in (5, 300, 4000, 70000):
        per = rng.integers(0, 1 << 20, 365)
        ints = ((per[np.arange(n) % 365] + rng.integers(-2, 3, n)) * 8 + rng.integers(0, 2, n)).astype(np.int64)
        u32s = ((per[np.arange(n) % 365] % 10000 + rng.integers(0, 3, n)) * 12).astype(np.uint32)
        cents = ((per[np.arange(n) % 365] % 9000 + 1000 + rng.integers(0, 3, n)) / 100.0)
        f32q = (per[np.arange(n) % 365] + rng.standard_normal(n)).astype(np.float32).astype(np.float64)
        for nums, kw in ((ints, dict(mode=4, mode_u64=8, delta=3)), (u32s, dict(mode=4, mode_u64=12, delta=3)),
                         (cents, dict(mode=2, mode_f64=0.01, delta=3)), (cents.astype(np.float32), dict(mode=2, mode_f64=0.01, delta=3)),
                         (f32q, dict(mode=3, mode_u64=29, delta=3))):
            want = O.simple_compress(nums, O.make_config(**kw))
            got = U.gpu_simple_compress(nums, G.make_config(**kw))
            assert got == want, (nums.dtype, n, kw)
            assert U.bits_equal(U.gpu_simple_decompress(got, nums.dtype, nums.size), nums), (nums.dtype, n, kw)


def test_auto_specs(L):
    """ModeSpec::Auto / DeltaSpec::Auto (the reference's default ChunkConfig, and the NULL config of its C ABI)."""
    rng = np.random.default_rng(31)
    n = 30000
    cases = {
        "ramp_u64": U.synth("c2", n),
        "decimals_f64": rng.integers(1000, 10000, n) / 100.0,
        "decimals_f32": (rng.integers(1000, 10000, n) / 100.0).astype(np.float32),
        "quantised_f64": (rng.standard_normal(n) * 100).astype(np.float32).astype(np.float64),
        "intmult_i64": (rng.integers(-1000, 1000, n) * 8 - 1).astype(np.int64),
        "walk_i32": np.cumsum(rng.integers(-50, 60, n)).astype(np.int32),
        "smooth_i64": (np.cumsum(np.cumsum(rng.integers(-3, 4, n))) + (1 << 30)).astype(np.int64),
        # high consecutive orders: the second wave of delta trials (orders 4..7) decides these
        "quintic_i64": (np.arange(n, dtype=np.int64) ** 5 % (1 << 62) + rng.integers(0, 2, n)).astype(np.int64),
        "order6_i64": np.cumsum(np.cumsum(np.cumsum(np.cumsum(np.cumsum(np.cumsum(rng.integers(-1, 2, n))))))).astype(np.int64),
        "order4_u32": (np.cumsum(np.cumsum(np.cumsum(np.cumsum(rng.integers(0, 3, n))))) % (1 << 32)).astype(np.uint32),
        "seasonal_i64": U.synth("c4", n),
        "uniform_u32": rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32),
        "normal_f32": rng.standard_normal(n).astype(np.float32),
        "tiny_u32": np.arange(9, dtype=np.uint32),
        "small_i64": np.arange(300, dtype=np.int64) ** 2,
    }
    for name, nums in cases.items():
        for kw in (dict(), dict(mode=1), dict(delta=1), dict(level=4)):
            want = O.simple_compress(nums, O.make_config(**kw))
            got = U.gpu_simple_compress(nums, G.make_config(**kw))
            assert got == want, (name, kw)
    # the reference C ABI with a NULL config: Auto/Auto, level 8, uniform dtype byte (pco_c/src/lib.rs:34-55, standalone/simple.rs:22-48)
    nums = cases["decimals_f64"]
    cap = L.pco_standalone_guarantee_file_size(nums.size, 6); dst = np.zeros(cap, np.uint8); w = C.c_size_t(0)
    G.check(L.pco_standalone_simple_compress_into(nums.ctypes.data_as(C.c_void_p), nums.size, 6, None, dst.ctypes.data_as(C.c_void_p), cap, C.byref(w)))
    assert bytes(dst[: w.value]) == O.simple_compress(nums, O.make_config(), uniform_type=True)
    import pcodec_amd as P  # default ChunkConfig() round trip through the Python mirror
    np.testing.assert_array_equal(P.standalone.simple_decompress(P.standalone.simple_compress(nums, P.ChunkConfig())), nums)


def test_auto_mode_device_screens_agree_with_the_oracle(L):
    """ModeSpec::Auto with part of the decision taken on the device (triple-GCD counts for ints, the float screen): data that is
    structured only in part -- a fraction of multiples / decimals / quantised numbers among noise -- sits on either side of every
    gate (sample too small, fewer than half with trailing zeros, too few converging pairs, no k saving bits)."""
    rng = np.random.default_rng(2024)
    bad = []
    for n in (40, 300, 5000, 30000):
        for frac in (0.0, 0.02, 0.2, 0.45, 0.5, 0.55, 0.8, 1.0):
            pick = rng.random(n) < frac
            noise64 = rng.standard_normal(n) * 100
            cases = {
                "f64_decimals": np.where(pick, np.round(noise64, 1), noise64),
                "f32_decimals": np.where(pick, np.round(noise64, 2), noise64).astype(np.float32),
                "f64_thirds": np.where(pick, rng.integers(-3000, 3000, n) / 3.0, noise64),
                "f32_pow2_steps": np.where(pick, rng.integers(-3000, 3000, n) * 0.125, noise64).astype(np.float32),
                "f64_quantised": np.where(pick, (noise64.view(np.uint64) & ~np.uint64(0xFFFFFF)).view(np.float64), noise64),
                "f32_quantised": np.where(pick, (noise64.astype(np.float32).view(np.uint32) & ~np.uint32(0x3FF)).view(np.float32), noise64.astype(np.float32)),
                "f32_specials": np.where(pick, np.float32(np.inf), noise64.astype(np.float32)),
                "f64_huge": np.where(pick, 1e308, noise64),
                "i64_multiples": np.where(pick, rng.integers(-10**6, 10**6, n) * 7919, rng.integers(-10**12, 10**12, n)).astype(np.int64),
                "u32_multiples": np.where(pick, rng.integers(0, 10**6, n) * 100, rng.integers(0, 1 << 32, n)).astype(np.uint32),
                "i16_multiples": np.where(pick, rng.integers(-1000, 1000, n) * 12, rng.integers(-(1 << 15), 1 << 15, n)).astype(np.int16),
                "u64_two_bases": np.where(pick, rng.integers(0, 10**6, n) * 6, rng.integers(0, 10**6, n) * 10).astype(np.uint64),
            }
            for name, nums in cases.items():
                want = O.simple_compress(nums, O.make_config(delta=1))
                got = U.gpu_simple_compress(nums, G.make_config(delta=1))
                if got != want: bad.append((name, n, frac))
    # full-size chunks: 2^18 (a 6563-number sample, decided on the device) and 300 000 (7509: beyond the device kernels' capacity, host path),
    # both specs Auto
    for n in (1 << 18, 300000):
        noise = rng.standard_normal(n) * 100
        for name, nums in (("f64_decimals", np.round(noise, 2)), ("f32_eighths", (rng.integers(-30000, 30000, n) * 0.125).astype(np.float32)),
                           ("f64_noise", noise), ("i64_multiples", (rng.integers(-10**6, 10**6, n) * 7919).astype(np.int64)),
                           ("u32_noise", rng.integers(0, 1 << 32, n).astype(np.uint32))):
            kw = dict(max_page_n=1 << 19)
            want = O.simple_compress(nums, O.make_config(**kw))
            got = U.gpu_simple_compress(nums, G.make_config(**kw))
            if got != want: bad.append((name, n, "full size"))
    assert not bad, bad[:12]


def _pair_gcd(hi, lo, F, prec):
    """mode/float_mult.rs:101-142 in numpy scalars of type F: one IEEE operation per step, as the device does them"""
    tiny, eps, p16, p6 = F(2.0) ** F(-(prec - 6)), F(2.0) ** F(-prec), F(2.0) ** F(-16), F(64.0)
    if lo <= hi * tiny or lo == hi: return None
    gv, ge, lv, le = hi, F(0), lo, F(0)
    while True:
        prev = gv
        q = F(gv / lv)
        fl = F(np.floor(q))
        ratio = F(fl + F(1)) if F(q - fl) >= F(0.5) else fl        # round half away from zero (q > 0; q - floor(q) is exact)
        ge = F(ge + F(F(ratio * le) + F(gv * eps)))
        gv = F(abs(F(gv - F(ratio * lv))))
        if gv <= F(prev * p16) or gv <= ge: return lv
        if gv <= F(hi * tiny) or gv <= F(ge * p6): return None
        gv, lv = lv, gv; ge, le = le, ge


def test_auto_float_screen_matches_an_ieee_reference(L):
    """Stage 1 of Auto mode detection on floats (auto_float_stats_kernel): its sample filter, trailing-zeros histogram and power of two,
    the converging approximate-GCD pairs, the percentile similarity counts, the picked GCD and its centring against the same steps in
    numpy scalars (every operation rounded once, no contraction)."""
    import math
    L.pco_gfx_debug_float_screen.restype = C.c_int
    rng = np.random.default_rng(606)
    for dt, prec, dtype_id in ((np.float32, 23, G.DTYPE_BYTE["float32"]), (np.float64, 52, G.DTYPE_BYTE["float64"])):
        F = dt; bits_n = 32 if dt == np.float32 else 64; bias = 127 if dt == np.float32 else 1023
        for kind in range(7):
            n = 4000
            base = rng.standard_normal(n) * 50
            with np.errstate(all="ignore"):
                vals = [base, np.round(base, 1), rng.integers(-500, 500, n) * 0.3 + (rng.random(n) < 0.3) * rng.standard_normal(n) * 1e-3,
                        rng.integers(1, 2000, n) * 0.125, np.where(rng.random(n) < 0.1, np.nan, base * 1e30), rng.integers(1, 40, n) / 7.0,
                        rng.integers(1, 3000, n) * 0.01][kind].astype(dt)
            vals[::97] = dt(0.0); vals[5::131] = dt(np.inf); vals[7::113] = np.finfo(dt).tiny / dt(4)
            out = (C.c_uint32 * 80)()
            G.check(L.pco_gfx_debug_float_screen(vals.ctypes.data_as(C.c_void_p), C.c_size_t(n), C.c_uint32(dtype_id), out))
            with np.errstate(all="ignore"):
                a = np.abs(vals)
                keep = np.isfinite(a) & (a >= np.finfo(dt).tiny) & (a <= np.finfo(dt).max / dt(2))
            s = a[keep]
            ubits = [int(b) for b in s.view(np.uint32 if dt == np.float32 else np.uint64)]
            tz_raw = np.array([(b & -b).bit_length() - 1 for b in ubits])
            hist = np.bincount(np.minimum(tz_raw, prec), minlength=56)[:56]
            with np.errstate(all="ignore"):
                g = [_pair_gcd(max(s[i], s[i + 1]), min(s[i], s[i + 1]), F, prec) for i in range(0, len(s) - 1, 2)]
            g = sorted(x for x in g if x is not None)
            need = 1 + math.ceil(len(s) * 0.001); required = max(math.ceil(len(s) * 0.5), 10)
            sims, cands = [], []
            for pct in (0.1, 0.3, 0.5):
                c = g[int(pct * len(g))] if g else F(0)
                cands.append(c); sims.append(sum(1 for x in g if F(abs(F(x - c))) < F(F(0.01) * c)))
            assert out[0] == len(s), (dt, kind)
            assert out[1] == int((tz_raw >= 5).sum()), (dt, kind)
            assert list(out[6:62]) == list(hist), (dt, kind)
            assert out[2] == len(g), (dt, kind, out[2], len(g))
            assert list(out[3:6]) == sims, (dt, kind, list(out[3:6]), sims)
            # trailing zeros: the common power of two and how many numbers are whole multiples of it
            if out[1] >= required:
                exps = [(b >> prec) - bias for b in ubits]
                divp = [e - max(prec - int(t), 0) for e, t in zip(exps, tz_raw)]
                k = min(d for d, t in zip(divp, tz_raw) if t >= 5)
                assert np.int32(np.uint32(out[63])) == k, (dt, kind)
                assert out[64] == sum(1 for d, e in zip(divp, exps) if d >= k and e < k + bits_n), (dt, kind)
            # Euclid: the first percentile value enough GCDs agree with, centred
            has = len(g) >= need and any(x >= need for x in sims)
            assert out[62] == int(has), (dt, kind)
            if has:
                basev = cands[[x >= need for x in sims].index(True)]
                inv = F(F(1.0) / basev); tsum = F(0); tw = F(0)
                with np.errstate(all="ignore"):
                    for x in s:
                        q = F(x * inv); fl = F(np.floor(q)); mult = F(fl + F(1)) if F(q - fl) >= F(0.5) else fl
                        mb = int(np.array([mult]).view(np.uint32 if dt == np.float32 else np.uint64)[0]) & ((1 << (bits_n - 1)) - 1)
                        me = (mb >> prec) - bias
                        if 0 <= me < prec and mult != 0:
                            over = F(F(mult * basev) - x)
                            w = F(prec - me)
                            tsum = F(tsum + F(w * F(over / mult))); tw = F(tw + w)
                    want = F(basev - F(tsum / tw))
                got_bits = out[65] | (out[66] << 32)
                want_bits = int(np.array([want]).view(np.uint32 if dt == np.float32 else np.uint64)[0])
                assert got_bits == want_bits or (np.isnan(want) and True), (dt, kind, hex(got_bits), hex(want_bits))


def test_auto_mode_host_path_stays_in_step(L):
    """The host path of Auto mode detection (f16, samples beyond the device kernels' capacity, overflowing GCD lists) is the same code the
    A/B switch PCO_GFX_AUTO_MODE_ON_HOST forces for everything: under it the same inputs must give the same bytes."""
    import subprocess, sys
    code = r'''
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, gpu_util as U, oracle_lib as O
from pcodec_amd import _lib as G
rng = np.random.default_rng(77)
n = 40000
noise = rng.standard_normal(n) * 100
cases = [np.round(noise, 2), np.round(noise, 1).astype(np.float32), noise, (rng.integers(-10**6, 10**6, n) * 7919).astype(np.int64),
         rng.integers(0, 1 << 32, n).astype(np.uint32), (noise.view(np.uint64) & ~np.uint64(0xFFFFFF)).view(np.float64), (rng.integers(0, 2000, n) * np.float16(0.1)).astype(np.float16)]
for nums in cases:
    assert U.gpu_simple_compress(nums, G.make_config()) == O.simple_compress(nums, O.make_config()), nums.dtype
print("ok")
'''
    env = dict(os.environ, PCO_GFX_AUTO_MODE_ON_HOST="1")
    out = subprocess.run([sys.executable, "-c", code], cwd=os.path.join(HERE, ".."), env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_encode_matrix_small(L):
    rng = np.random.default_rng(99)
    bad = []
    for dt in (np.uint32, np.int32, np.uint64, np.int64, np.float32, np.float64, np.uint16, np.int16, np.uint8, np.int8):
        for n in (1, 2, 3, 255, 256, 257, 513, 2000):
            for dk, do in ((1, 0), (2, 1), (2, 3), (2, 7)):
                if np.dtype(dt).kind == "f":
                    nums = [(rng.standard_normal(n) * 50).astype(dt), (rng.integers(0, 30, n) * 0.25).astype(dt)][n % 2]
                    nums[rng.integers(0, n)] = np.nan
                else:
                    ii = np.iinfo(dt)
                    nums = [rng.integers(max(ii.min, -(1 << 40)), min(ii.max, 1 << 40), n), np.cumsum(rng.integers(-3, 9, n)) % min(ii.max, 1 << 40)][n % 2].astype(dt)
                kw = dict(mode=1, delta=dk, delta_order=do)
                want = O.simple_compress(nums, O.make_config(enable_8_bit=True, **kw))
                got = U.gpu_simple_compress(nums, G.make_config(enable_8_bit=True, **kw))
                if got != want:
                    bad.append((np.dtype(dt).name, n, dk, do))
                elif not U.bits_equal(U.gpu_simple_decompress(got, nums.dtype, n), nums):
                    bad.append(("decode", np.dtype(dt).name, n, dk, do))
    assert not bad, bad[:10]


def test_eight_bit_types(L):
    """u8 / i8 (chunk_config.rs:306: refused unless enable_8_bit): the reference's v1_0_0_{u8,i8}.pco golden files are
    reproduced byte for byte by the GPU encoder with Auto mode + Auto delta, and decode to the generator's arrays."""
    from test_oracle_golden import EXPECTED, asset
    for name in ("v1_0_0_u8.pco", "v1_0_0_i8.pco"):
        exp = EXPECTED[name]
        assert U.bits_equal(U.gpu_simple_decompress(asset(name), exp.dtype, exp.size), exp)
        assert U.gpu_simple_compress(exp, G.make_config(enable_8_bit=True)) == bytes(asset(name)), name
    with pytest.raises(G.PcoGfxError) as ei:   # refused without the opt-in, like the reference
        U.gpu_simple_compress(np.arange(100, dtype=np.uint8), G.make_config(mode=1, delta=1))
    assert ei.value.status == G.ST_INVALID_ARGUMENT
    rng = np.random.default_rng(12)
    for dt in (np.uint8, np.int8):
        nums = (np.cumsum(rng.integers(-2, 3, 70000)) % 200).astype(dt)
        for kw in (dict(mode=1, delta=1), dict(mode=1, delta=2, delta_order=1), dict(mode=4, mode_u64=5, delta=1), dict()):
            want = O.simple_compress(nums, O.make_config(enable_8_bit=True, **kw))
            got = U.gpu_simple_compress(nums, G.make_config(enable_8_bit=True, **kw))
            assert got == want, (dt, kw)
            assert U.bits_equal(U.gpu_simple_decompress(got, nums.dtype, nums.size), nums)


def test_levels_and_histogram_paths(L):
    """compression levels 0..8 and both histogram paths (value-space counting vs radix sort), incl. heavy ties."""
    rng = np.random.default_rng(5)
    n = 20000
    datasets = {
        "narrow": rng.integers(0, 300, n).astype(np.uint64),
        "wide": rng.integers(0, 1 << 62, n, dtype=np.uint64),
        "ties": np.where(rng.random(n) < 0.6, 7, rng.integers(0, 1 << 40, n)).astype(np.uint64),
        "geometric": (rng.geometric(0.05, n) * 1000003).astype(np.uint32),
        "normal_f32": rng.standard_normal(n).astype(np.float32),
        "lomax_i32": (rng.pareto(0.5, n) * 10).clip(0, 2e9).astype(np.int32),
    }
    for level in (0, 1, 4, 8, 9, 10, 12):   # 9..12: more than 256 histogram bins at n >= 2^13 (sort histogram, block-wide DP, big page tables)
        for name, nums in datasets.items():
            kw = dict(level=level, mode=1, delta=1)
            want = O.simple_compress(nums, O.make_config(**kw))
            _, _, fb = O.chunk_plan(nums, O.make_config(**kw))
            got = U.gpu_simple_compress(nums, G.make_config(**kw))
            if fb:  # the reference's order-dependent heapsort fallback ran: bytes may legitimately differ
                assert U.bits_equal(O.simple_decompress(got, nums.dtype, cap=n + 8), nums)
            else:
                assert got == want, (level, name)


def test_histogram_range_boundaries_and_skew(L):
    """The three histogram kernels split on the value range (max - min): < 4096 LDS counting, < 32768 wide LDS counting,
    else bucket pre-selection + radix sort of the needed buckets.  Ranges straddling both boundaries, heavy skew (a dense
    core plus far outliers: every queried rank in one bucket) and ties must all give the oracle's bytes."""
    rng = np.random.default_rng(21)
    n = 70000
    cases = {}
    for r in (4094, 4095, 4096, 4097, 16382, 16383, 16384, 16385, 32766, 32767, 32768, 32769, 100000):
        x = rng.integers(0, r + 1, n).astype(np.uint64); x[0] = 0; x[1] = r      # range exactly r
        cases[f"uniform_range_{r}"] = x + np.uint64(1 << 33)
    core = rng.integers(1000, 1008, n).astype(np.int64)
    core[rng.integers(0, n, 20)] = rng.integers(-(1 << 50), 1 << 50, 20)
    cases["dense_core_with_outliers"] = core
    cases["two_clusters"] = np.where(rng.random(n) < 0.5, rng.integers(0, 50, n), rng.integers(1 << 40, (1 << 40) + 50, n)).astype(np.uint64)
    cases["random_u32"] = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    cases["random_f64"] = rng.standard_normal(n) * 1e6
    cases["wide_with_ties"] = (rng.integers(0, 300, n) * np.uint64(0x10000000001)).astype(np.uint64)
    for name, nums in cases.items():
        for kw in (dict(mode=1, delta=1), dict(mode=1, delta=2, delta_order=1)):
            ocfg = O.make_config(**kw)
            want = O.simple_compress(nums, ocfg)
            _, _, fb = O.chunk_plan(nums, ocfg)
            got = U.gpu_simple_compress(nums, G.make_config(**kw))
            if fb:  # the reference's order-dependent heapsort fallback ran
                assert U.bits_equal(O.simple_decompress(got, nums.dtype, cap=n + 8), nums), name
            else:
                assert got == want, (name, kw)
            assert U.bits_equal(U.gpu_simple_decompress(got, nums.dtype, n), nums), name


def test_flat_bucket_map_of_the_select_histogram(L):
    """enc_hist_select_kernel takes equal power-of-two segments instead of sample quantiles when the sorted sample says the data is spread
    evenly (no heavy value, no segment above three times its share, at least 90 segments): the decision's edges -- ranges that make 89 / 90 /
    127 / 128 segments, 16-bit and 64-bit types --, unstored positions (delta, several pages), and data that only LOOKS flat to the sample (a
    heavy value at positions the 2048 evenly spaced sample positions miss: one bucket holds thousands of latents) all give the oracle's bytes."""
    rng = np.random.default_rng(2206)
    n = 70000
    cases = {}
    for bl, segs in ((20, 89), (20, 90), (20, 127), (20, 128), (40, 100), (63, 128)):
        r = (segs << (bl - 7)) - 1 if segs < 128 else (1 << bl) - 1     # range with bit length bl whose top seven bits + 1 make `segs` segments
        x = rng.integers(0, r + 1, n, dtype=np.uint64); x[0] = 0; x[1] = r
        cases[f"u64_bl{bl}_segs{segs}"] = x + np.uint64(12345)
    cases["u16_full"] = rng.integers(0, 1 << 16, n, dtype=np.uint64).astype(np.uint16)
    cases["i16_most"] = rng.integers(-30000, 30000, n).astype(np.int16)
    cases["u32_full"] = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    hidden = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    pos = np.arange(n); sampled = set(((np.arange(2048, dtype=np.uint64) * np.uint64(n)) // np.uint64(2048)).tolist())
    free = np.array([p for p in pos if p not in sampled])
    hidden[rng.choice(free, 6000, replace=False)] = 0x12345678
    cases["heavy_value_the_sample_misses"] = hidden
    for name, nums in cases.items():
        for kw in (dict(mode=1, delta=1), dict(mode=1, delta=2, delta_order=2), dict(mode=1, delta=2, delta_order=1, max_page_n=9000)):
            ocfg = O.make_config(**kw)
            want = O.simple_compress(nums, ocfg)
            _, _, fb = O.chunk_plan(nums, ocfg)
            got = U.gpu_simple_compress(nums, G.make_config(**kw))
            if fb:  # the reference's order-dependent heapsort fallback ran
                assert U.bits_equal(O.simple_decompress(got, nums.dtype, cap=n + 8), nums), (name, kw)
            else:
                assert got == want, (name, kw)
            assert U.bits_equal(U.gpu_simple_decompress(got, nums.dtype, n), nums), (name, kw)


def test_multi_page_chunks_match_the_oracle_bytes(L):
    """PagingSpec::EqualPagesUpTo (chunk_config.rs:134-182): a standalone file is always one page per chunk, so the
    wrapped surface is compared page by page with the oracle's wrapped chunk (meta + pages)."""
    rng = np.random.default_rng(22)
    nums = (np.cumsum(rng.integers(-20, 90, 50000)) + (1 << 35)).astype(np.int64)
    for kw in (dict(mode=1, delta=2, delta_order=1, max_page_n=7000), dict(mode=1, delta=1, max_page_n=4096), dict(mode=4, mode_u64=3, delta=2, delta_order=2, max_page_n=20000)):
        cfg = G.make_config(**kw)
        cc = C.c_void_p()
        G.check(L.pco_chunk_compressor_new(nums.ctypes.data_as(C.c_void_p), C.c_size_t(nums.size), C.c_ubyte(4), C.byref(cfg), C.byref(cc)))
        L.pco_chunk_compressor_n_pages.restype = C.c_size_t; L.pco_chunk_compressor_page_n.restype = C.c_size_t
        n_pages = L.pco_chunk_compressor_n_pages(cc)
        buf = np.zeros(1 << 20, np.uint8); w = C.c_size_t(0)
        G.check(L.pco_chunk_compressor_write_meta(cc, buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size), C.byref(w)))
        meta = bytes(buf[: w.value]); pages = []; page_ns = []
        for i in range(n_pages):
            G.check(L.pco_chunk_compressor_write_page(cc, C.c_size_t(i), buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size), C.byref(w)))
            pages.append(bytes(buf[: w.value])); page_ns.append(L.pco_chunk_compressor_page_n(cc, C.c_size_t(i)))
        L.pco_chunk_compressor_free(cc)
        want_meta, want_pages, want_ns = O.wrapped_compress(nums, O.make_config(**kw))
        assert page_ns == want_ns and meta == want_meta, kw
        assert pages == want_pages, kw


# what the sweep may leave uncompared, in total and by reason (anything else fails the test): the reference's order-dependent
# histogram branch and nothing the product refuses
SWEEP_SKIP_BUDGET = 0.03


def check_skips(skipped, n_cases):
    total = sum(skipped.values())
    allowed = {k: v for k, v in skipped.items() if k.startswith("bytes not compared: reference heapsort-fallback") or k.startswith("oracle refused")}
    assert allowed == skipped, f"the sweep skipped cases for reasons outside the budget: {skipped}"
    assert total <= max(2, int(SWEEP_SKIP_BUDGET * n_cases)), f"skip budget exceeded: {skipped}"


def test_randomised_parity_sweep(L):
    """Random dtype x size x distribution x ChunkConfig (tests/fuzz_util.py): encode bytes identical to the oracle's, decode of
    the oracle's bytes bit-exact.  Sizes include n = 1 (mod 256), 2^18 and 2^18 + 1.  The number of cases that were not compared
    is bounded and their reasons are listed: refusing more inputs does not pass.
    (This sweep found the 1-byte case missing from the Auto-delta sample gather, and the encode walker's divergent wave vote.)"""
    import fuzz_util
    bad, skipped, _ = fuzz_util.run(400, 2024, max_level=12)
    assert not bad, bad[:10]
    check_skips(skipped, 400)
    bad, skipped, _ = fuzz_util.run(250, 2025, only_8bit=True)
    assert not bad, bad[:10]
    check_skips(skipped, 250)


def test_randomised_batched_sweep(L):
    """The same through the batched device API: up to 40 chunks of mixed dtypes and sizes per call, one config per call."""
    import fuzz_util
    bad, skipped = fuzz_util.run_batched(12, 77, max_level=12)
    assert not bad, bad[:10]
    assert not skipped, skipped


def test_two_variable_chunks_with_unequal_batch_counts(L):
    """Round 1's open encoder bug: in enc_walk_kernel idle quads (a trivial variable's slot) left the wave through a vote taken
    inside a divergent branch, after which the butterfly maximum over the quads' batch counts read dead lanes; a wave whose
    first live quad owned fewer full batches than a later one (delta'd primary with n - state latents next to a secondary with
    n, n = 1 mod 256) then stopped walking early.  Every (two-variable mode) x (delta) x (n around multiples of 256) x
    (which variable is trivial), plus the sweep's original failing input."""
    rng = np.random.default_rng(2025)
    d = np.load(os.path.join(HERE, "golden", "fuzz_case_2025_188.npz"))["nums"]
    cases = [("sweep 2025/188", np.resize(d, n).astype(dt), dict(level=4, mode=4, mode_u64=134, delta=3))
             for n in (257, 513, 1025) for dt in (np.uint8, np.uint16, np.uint32, np.uint64, np.int32)]
    deltas = (dict(delta=1), dict(delta=2, delta_order=1), dict(delta=2, delta_order=3), dict(delta=3))
    for n in (1, 2, 256, 257, 258, 260, 513, 769, 1025, 4097, (1 << 16) + 1, (1 << 18) + 1):
        step = np.repeat(rng.integers(0, 2, -(-n // 40)), 40)[:n]           # primary: long constant stretches (trivial lookbacks)
        noisy = rng.integers(0, 6, n)
        ints = {"noisy secondary": (1000 + step) * 8 + noisy, "trivial secondary": (np.cumsum(rng.integers(0, 5, n)) + 7) * 8 + 3,
                "trivial primary": 77 * 8 + noisy}
        for tag, x in ints.items():
            for dk in deltas:
                cases.append((f"int-mult {tag}", x.astype(np.uint32), dict(mode=4, mode_u64=8, **dk)))
        cents = {"noisy secondary": (1000 + step) * 0.01 + (noisy - 3) * 1e-13, "trivial secondary": np.round(rng.integers(100, 9000, n)) / 100.0,
                 "trivial primary": 7.25 + (noisy - 3) * 1e-15}
        for tag, x in cents.items():
            for dk in deltas:
                cases.append((f"float-mult {tag}", x.astype(np.float64), dict(mode=2, mode_f64=0.01, **dk)))
        q = {"noisy secondary": ((1000.0 + step).astype(np.float32).view(np.uint32) + noisy.astype(np.uint32)).view(np.float32),
             "trivial secondary": (rng.integers(0, 50, n) * 4.0).astype(np.float32), "trivial primary": (np.float32(3.0).view(np.uint32) + noisy.astype(np.uint32)).view(np.float32)}
        for tag, x in q.items():
            for dk in deltas:
                cases.append((f"float-quant {tag}", np.ascontiguousarray(x), dict(mode=3, mode_u64=12, **dk)))
    bad = []
    for tag, nums, kw in cases:
        want = O.simple_compress(nums, O.make_config(enable_8_bit=True, **kw))
        got = U.gpu_simple_compress(nums, G.make_config(enable_8_bit=True, **kw))
        if got != want: bad.append(("bytes", tag, nums.dtype.name, nums.size, kw))
        elif not U.bits_equal(U.gpu_simple_decompress(got, nums.dtype, nums.size), nums): bad.append(("decode", tag, nums.dtype.name, nums.size, kw))
    assert not bad, (len(bad), bad[:8])
    # the same shapes packed into one batched call: waves of the walker then hold quads with different batch counts
    arrays = [c[1] for c in cases if c[2].get("mode") == 4 and c[2].get("delta") == 3 and c[1].dtype == np.uint32][:40]
    chunks, back = U.gpu_batched(arrays, G.make_config(mode=4, mode_u64=8, delta=3))
    for a, ch, b in zip(arrays, chunks, back):
        want = O.simple_compress(a, O.make_config(mode=4, mode_u64=8, delta=3, max_page_n=max(a.size, 1 << 18)))   # a batched task is ONE chunk
        assert ch == U.chunk_of_file(want, len(ch)) and U.bits_equal(a, b), a.size


def test_half_precision_floats(L):
    """f16 (data_types/float.rs:254-366): Classic, FloatQuant, explicit FloatMult bases (arithmetic through f32, one rounding per operation) and
    ModeSpec::Auto, on all 65536 bit patterns (NaN payloads, infinities, subnormals, signed zeros) and on structured data; also exhaustively:
    every f16 value joined back from its own (mult, adjustment) pair."""
    rng = np.random.default_rng(77)
    every = np.arange(65536, dtype=np.uint16).view(np.float16)
    n = 40000
    with np.errstate(all="ignore"):
        cases = {
            "every_pattern": every,
            "every_pattern_shuffled": rng.permutation(every),
            "tenths": (rng.integers(0, 2000, n) * np.float16(0.1)).astype(np.float16),
            "hundredths": (rng.integers(0, 500, n).astype(np.float32) * 0.01).astype(np.float16),
            "integers": rng.integers(-2000, 2000, n).astype(np.float16),
            "past_2048": rng.integers(-60000, 60000, n).astype(np.float16),      # beyond the greatest precise integer: the latent continues in bit steps
            "quantised": (rng.normal(0, 100, n).astype(np.float16).view(np.uint16) & 0xFFE0).view(np.float16),
            "normal": rng.normal(0, 1, n).astype(np.float16),
            "sevenths": (rng.integers(1, 60, n).astype(np.float32) * 0.3).astype(np.float16),
            "tiny": np.float16([0.5, -0.0, np.nan, np.inf, 6e-8, 65504, -65504]),
        }
    specs = [dict(mode=1), dict(), dict(mode=3, mode_u64=3), dict(mode=3, mode_u64=10)]
    specs += [dict(mode=2, mode_f64=b) for b in (0.1, 0.01, 0.25, 3.0, 1.0 / 3.0, -0.3, 1e-7, 6e-8, 65504.0, 1000.0)]
    bad = []
    for name, nums in cases.items():
        for kw in specs:
            for dk in (dict(delta=1), dict(delta=2, delta_order=1), dict()):
                okw = dict(kw); okw.update(dk)
                want = O.simple_compress(nums, O.make_config(**okw))
                got = U.gpu_simple_compress(nums, G.make_config(**okw))
                if got != want: bad.append((name, okw)); continue
                if not U.bits_equal(U.gpu_simple_decompress(got, np.float16, nums.size), nums): bad.append(("decode", name, okw))
    assert not bad, bad[:10]
    for b in (0.0, float("inf"), float("nan"), 1e-9, 1e6):      # a base that is zero / not finite once rounded to f16: refused like the reference
        with pytest.raises(G.PcoGfxError) as ei:
            U.gpu_simple_compress(cases["tenths"], G.make_config(mode=2, mode_f64=b, delta=1))
        assert ei.value.status == G.ST_INVALID_ARGUMENT, b
        with pytest.raises(O.OracleError):
            O.simple_compress(cases["tenths"], O.make_config(mode=2, mode_f64=b, delta=1))


def test_unsupported_requests_fail_loudly(L):
    nums = np.arange(1000, dtype=np.uint32)
    for kw in (dict(mode=5, delta=1), dict(mode=1, delta=4, delta_order=2)):   # Dict / Conv1 ENCODE: outside the hot path
        with pytest.raises(G.PcoGfxError) as ei:
            U.gpu_simple_compress(np.tile(nums, 300), G.make_config(**kw))
        assert ei.value.status in (G.ST_UNSUPPORTED, G.ST_INVALID_ARGUMENT)
    with pytest.raises(G.PcoGfxError) as ei:
        U.gpu_simple_compress(nums, G.make_config(level=13, mode=1, delta=1))
    assert ei.value.status == G.ST_INVALID_ARGUMENT
    with pytest.raises(G.PcoGfxError):
        U.gpu_simple_compress(nums, G.make_config(mode=2, mode_f64=0.1, delta=1))  # float mode on ints


def test_decode_expanders_under_the_walk(L):
    """decode_trail.hip: the chunks of one latent variable with 2..64 bins and offsets of up to 16 bits are expanded on a second stream
    while the tANS walk runs, eight chunks per walker block, two per expander wave.  One batched decode call whose walker blocks mix
    such chunks with every kind the expanders must leave alone (two variables, lookback, one bin, wide offsets, more than 64 bins),
    with short and ragged chunks (fewer than two batches, a last batch of one number, delta orders 0..3 so that the tail batches hold
    only delta state), more chunks than one round of walker blocks, and a damaged chunk in the middle of a block -- every number must come
    back, the damaged chunk must report its error and nothing else may be disturbed.  Decoded twice: with the expanders and with
    PCO_GFX_DEC_TRAIL=0 semantics (the same library cannot switch in-process: the reference result is the oracle's decoder)."""
    import torch
    rng = np.random.default_rng(77)
    L_ = G.lib()
    arrays = []; cfgs = []
    def ramp(n, dt, step=1000, noise=512):
        return (np.arange(n, dtype=np.int64) * step + rng.integers(0, noise, n) + (1 << 30)).astype(dt)
    sizes = [700, 2, 255, 256, 257, 511, 513, 1000, 4096, 4097, 70000, 3, 258, 769, 1]
    for rep in range(64):
        n = sizes[rep % len(sizes)]
        kind = rep % 16
        if kind == 0: a, kw = ramp(n, np.uint64), dict(mode=1, delta=2, delta_order=1)
        elif kind == 1: a, kw = ramp(n, np.int32, 7, 40), dict(mode=1, delta=2, delta_order=2)
        elif kind == 2: a, kw = ramp(n, np.uint32, 3, 9), dict(mode=1, delta=2, delta_order=3)                       # order 3: left to dec_expand_kernel
        elif kind == 3: a, kw = (rng.integers(1000, 10000, n) / 100.0), dict(mode=2, mode_f64=0.01, delta=1)          # two variables
        elif kind == 4: a, kw = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32), dict(mode=1, delta=1) # one bin, 32-bit offsets
        elif kind == 5: a, kw = (rng.integers(-(1 << 40), 1 << 40, 365)[np.arange(n) % 365] + rng.integers(-3, 4, n)).astype(np.int64), dict(mode=1, delta=3)   # lookback
        elif kind == 6: a, kw = (rng.pareto(0.5, n) * 10).clip(0, 2e9).astype(np.int32), dict(mode=1, delta=1)         # many bins, wide offsets
        elif kind == 7: a, kw = rng.integers(0, 40, n).astype(np.uint16), dict(mode=1, delta=1)                        # a few bins, no offsets at all
        elif kind == 8: a, kw = ramp(n, np.int64, 5, 100), dict(mode=1, delta=1)                                       # no delta, narrow offsets
        elif kind == 9: a, kw = rng.standard_normal(n).astype(np.float32), dict(mode=1, delta=2, delta_order=1)
        # two latent variables: the expanders of the second kind (dec_trail_kernel<L, true>)
        elif kind == 10: a, kw = (np.cumsum(rng.integers(-50, 60, n)) / 100.0 + 500.0), dict(mode=2, mode_f64=0.01, delta=2, delta_order=1)   # float-mult, delta'd multiples
        elif kind == 11: a, kw = (rng.integers(0, 5000, n).astype(np.int64) * 1000 + rng.integers(0, 3, n)), dict(mode=4, mode_u64=1000, delta=1)   # int-mult with remainders
        elif kind == 12: a, kw = (rng.integers(8, 16, n) * 2.0 ** rng.integers(-3, 4, n)).astype(np.float32), dict(mode=3, mode_u64=20, delta=1)   # float-quant
        elif kind == 13: a, kw = ((rng.integers(10, 4000, n) * 0.1).astype(np.float32) * (1 + rng.integers(-2, 3, n).astype(np.float32) * np.float32(2 ** -22))), dict(mode=2, mode_f64=0.1, delta=1)   # f32 float-mult, adjustments of a few ulps
        elif kind == 14: a, kw = (rng.integers(-300, 300, n).astype(np.int32) * 7), dict(mode=4, mode_u64=7, delta=2, delta_order=2)   # int-mult, order 2, secondary constant
        else: a, kw = (rng.integers(1, 1 << 40, n).astype(np.float64) * 0.25), dict(mode=2, mode_f64=0.25, delta=1)   # float-mult whose multiples need offsets beyond 16 bits: left to dec_expand_kernel
        arrays.append(np.ascontiguousarray(a)); cfgs.append(kw)
    blobs = [U.chunk_of_file(f, len(f) - O_header_len(f) - 1) for f in (O.simple_compress(a, O.make_config(**kw)) for a, kw in zip(arrays, cfgs))]
    # many copies, so that the call needs more than one round of walker blocks per width; one damaged chunk per width in the middle
    reps = 40   # (the expanders take calls of at least 1024 chunks per number width: 28 of the 64 arrays are 64-bit, 32 are 32-bit; the four 16-bit ones stay below and walk, then expand)
    srcs, tasks_spec, want = [], [], []
    for r in range(reps):
        for i, (a, b) in enumerate(zip(arrays, blobs)):
            damaged = (r == reps // 2 and i % 16 in (0, 9, 10, 11) and len(b) > 64)
            bb = b[: len(b) // 2] if damaged else b
            srcs.append(bb); tasks_spec.append((a.dtype, a.size, damaged)); want.append(a)
    k = len(srcs)
    d_src = [torch.from_numpy(np.frombuffer(bb + b"\0" * 16, np.uint8).copy()).cuda() for bb in srcs]
    outs = [torch.full((max(a.nbytes, 1) + 16,), 0xAB, dtype=torch.uint8, device="cuda") for a in want]
    dtasks = (G.DecodeTask * k)(*[G.DecodeTask(d_src[i].data_ptr(), len(srcs[i]), outs[i].data_ptr(), tasks_spec[i][1], G.DTYPE_BYTE[np.dtype(tasks_spec[i][0]).name], 0) for i in range(k)])
    dres = (G.TaskResult * k)()
    rc = L_.pco_gfx_decompress_chunks(k, dtasks, dres, None, None)
    assert rc != 0   # (the damaged chunks)
    n_bad = 0
    for i in range(k):
        dt, n, damaged = tasks_spec[i]
        if damaged:
            assert dres[i].status == G.ST_INSUFFICIENT_DATA, (i, dres[i].status); n_bad += 1
            continue
        assert dres[i].status == 0 and dres[i].n_out == n and dres[i].consumed == len(srcs[i]), (i, dres[i].status, dres[i].n_out, n)
        got = outs[i][: want[i].nbytes].cpu().numpy().view(dt)
        assert U.bits_equal(got, want[i]), (i, cfgs[i % len(arrays)], n)
        assert bool((outs[i][want[i].nbytes:] == 0xAB).all()), i   # nothing written past the chunk's numbers
    assert n_bad >= 2
    # more walker blocks than the persistent expander grid holds (4 blocks per CU): the expander blocks take a second walker block each
    # (4096 u64 of a noisy ramp, delta order 1: five bins, 8-bit offsets -- a chunk the expanders take.  Round 4 used the 700-number chunk of the
    #  mix here, whose single bin makes the walker leave it to dec_expand_kernel: pco_gfx_trail_marked() showed that nothing of that call ever
    #  reached the expanders)
    a = ramp(4096, np.uint64); f2 = O.simple_compress(a, O.make_config(mode=1, delta=2, delta_order=1)); b = U.chunk_of_file(f2, len(f2) - O_header_len(f2) - 1)
    assert O.inspect_first_chunk(f2)[0].n_bins[1] > 1
    marked_before = L_.pco_gfx_trail_marked()
    k2 = 9000
    src = torch.from_numpy(np.frombuffer(b + b"\0" * 16, np.uint8).copy()).cuda()
    out = torch.zeros((k2, max(a.nbytes, 8)), dtype=torch.uint8, device="cuda")
    dt2 = (G.DecodeTask * k2)(*[G.DecodeTask(src.data_ptr(), len(b), out[i].data_ptr(), a.size, G.DTYPE_BYTE[a.dtype.name], 0) for i in range(k2)])
    dr2 = (G.TaskResult * k2)()
    G.check(L_.pco_gfx_decompress_chunks(k2, dt2, dr2, None, None))
    host = out.cpu().numpy()
    assert all(dr2[i].n_out == a.size for i in range(k2))
    assert (host[:, : a.nbytes] == a.view(np.uint8).reshape(1, -1)).all()
    assert L_.pco_gfx_trail_marked() - marked_before == k2   # every one of them was marked for the expanders under the walk


def O_header_len(f):
    bits = int.from_bytes(f[6:16], "little")
    return 6 + (6 + 1 + (bits & 63) + 7) // 8 + 2


def test_expanders_of_two_variable_chunks_with_a_constant_secondary(L):
    """dec_trail_kernel<L, true>: walker blocks full of chunks whose secondary variable is one bin without offset bits (exact decimals under
    float-mult, multiples under int-mult) take the interleaved pair path with nothing of the secondary fetched (trail_fast_pair2t); chunks
    with a real secondary in the same call take the two-unpacking path.  1100+ chunks of one width (the expanders run from 1024 on), ragged
    lengths (partial last batches, fewer than two batches, delta state in the tail), delta orders 0..2: every number comes back, the chunks
    equal the oracle's on a spread."""
    rng = np.random.default_rng(661)
    sizes = [4096, 1, 4097, 300, 70000, 511, 8192, 257, 1024, 12289]
    arrays = []
    def make(i, n):
        k = (i // 16) % 4    # runs of sixteen chunks of a kind: walker blocks (eight slots) and expander waves (two) of one kind, and mixed ones at the seams
        if k == 0: return rng.integers(1000, 900000, n) / 100.0                                   # decimals: constant adjustment
        if k == 1: return (np.cumsum(rng.integers(-50, 60, n)) + 100000) / 100.0                  # a walk in cents
        if k == 2: return rng.integers(1000, 900000, n) / 100.0 + (rng.integers(0, 3, n) - 1) * 1e-9   # a real secondary
        return rng.integers(1, 1 << 20, n) * 0.01
    for i in range(1152):
        arrays.append(np.ascontiguousarray(make(i, sizes[i % len(sizes)]), dtype=np.float64))
    for kw in (dict(mode=2, mode_f64=0.01, delta=1), dict(mode=2, mode_f64=0.01, delta=2, delta_order=1), dict(mode=2, mode_f64=0.01, delta=2, delta_order=2)):
        chunks, back = U.gpu_batched(arrays, G.make_config(**kw))
        for i, a in enumerate(arrays):
            assert U.bits_equal(back[i], a), (kw, i, a.size)
        for i in range(0, len(arrays), 37):
            want = O.simple_compress(arrays[i], O.make_config(**kw))
            assert chunks[i] == U.chunk_of_file(want, len(chunks[i])), (kw, i)
    ints = [np.ascontiguousarray((rng.integers(-3000, 3000, sizes[i % len(sizes)]) * 7).astype(np.int64)) for i in range(1100)]
    kw = dict(mode=4, mode_u64=7, delta=2, delta_order=1)
    chunks, back = U.gpu_batched(ints, G.make_config(**kw))
    for i, a in enumerate(ints):
        assert U.bits_equal(back[i], a), (kw, i, a.size)


def test_the_heapsort_branch_of_the_reference_histogram(L):
    """The one documented divergence, pinned by an input (histograms.rs:248-258).  On the two adversarial orders of
    tests/golden/hist_fallback.npz the reference's histogram heapsorts and applies apply_sorted's tie rule; the GPU computes the quickselect
    path's result as a function of the sorted numbers (the multiset rule).  What the GPU writes then is EXACTLY what the reference would
    write with that rule in the histogram and everything else unchanged (the oracle's test hook) -- it differs from the reference's bytes
    in the bins of ChunkMeta and what follows from them -- and either stream decodes to the input, on the GPU and in the oracle.
    On the same numbers in sorted order the branch does not run and the GPU's bytes are the reference's."""
    fx = np.load(os.path.join(HERE, "golden", "hist_fallback.npz"))
    kw = dict(mode=1, delta=1)
    for key in ("n5000", "n262144"):
        x = fx[key]
        literal = O.simple_compress(x, O.make_config(**kw))
        O.set_hist_rule(1)
        try:
            multiset = O.simple_compress(x, O.make_config(**kw))
        finally:
            O.set_hist_rule(0)
        _, _, fb = O.chunk_plan(x, O.make_config(**kw))
        assert fb and literal != multiset, key
        got = U.gpu_simple_compress(x, G.make_config(**kw))
        assert got == multiset, key
        info_l, bins_l = O.inspect_first_chunk(literal); info_g, bins_g = O.inspect_first_chunk(got)
        assert info_l.n_bins[1] != info_g.n_bins[1] or not np.array_equal(bins_l[1], bins_g[1])   # where the two part: the bins
        for blob in (got, literal):
            assert U.bits_equal(U.gpu_simple_decompress(blob, np.uint32, x.size), x)
            assert np.array_equal(O.simple_decompress(blob, np.uint32), x)
        xs = np.sort(x)
        assert U.gpu_simple_compress(xs, G.make_config(**kw)) == O.simple_compress(xs, O.make_config(**kw))


# ------------------------------------------------------------------------------------------------ errors
def test_truncation_and_corruption_are_reported(L):
    nums = np.array([0] * 50 + [1000] * 50, np.uint32)
    enc = O.simple_compress(nums, O.make_config(mode=1, delta=1))
    for i in range(len(enc)):  # tests/stability.rs:8-34 -- every strict prefix, the one that only lost the terminator byte included
        with pytest.raises(G.PcoGfxError) as ei:
            U.gpu_simple_decompress(enc[:i], np.uint32, 128)
        assert ei.value.status == G.ST_INSUFFICIENT_DATA, i
    # a file of several chunks that ends right behind its last chunk (standalone/decompressor.rs:190-200: chunk_preamble wants a byte)
    two = U.synth("c2", (1 << 18) + 700)
    enc2 = O.simple_compress(two, O.make_config(mode=1, delta=2, delta_order=1))
    assert enc2[-1] == 0
    assert U.bits_equal(U.gpu_simple_decompress(enc2, np.uint64, two.size), two)
    for cut in (1, 2):
        with pytest.raises(G.PcoGfxError) as ei:
            U.gpu_simple_decompress(enc2[:-cut], np.uint64, two.size)
        assert ei.value.status == G.ST_INSUFFICIENT_DATA, cut
    big = O.simple_compress(U.synth("c2", 5000), O.make_config(mode=1, delta=2, delta_order=1))
    rng = np.random.default_rng(1)
    for _ in range(200):  # tests/corruption.rs: never crash / hang; error or garbage, like the reference
        b = bytearray(big); b[rng.integers(0, len(b))] ^= 1 << rng.integers(0, 8)
        try:
            U.gpu_simple_decompress(bytes(b), np.uint64, 5000)
        except G.PcoGfxError as e:
            assert e.status in (G.ST_CORRUPTION, G.ST_INSUFFICIENT_DATA, G.ST_INVALID_ARGUMENT, G.ST_UNSUPPORTED)
    with pytest.raises(G.PcoGfxError):  # dst too small (pco_c/src/lib.rs:110-112)
        U.gpu_simple_decompress(big, np.uint64, 4999)
    with pytest.raises(G.PcoGfxError) as ei:  # wrong dtype
        U.gpu_simple_decompress(big, np.int64, 5000)
    assert ei.value.status == G.ST_CORRUPTION


# ------------------------------------------------------------------------------------------------ batched + wrapped
def test_batched_device_api_mixed_dtypes(L):
    import torch
    rng = np.random.default_rng(3)
    arrays = []
    for i in range(48):
        n = int(rng.integers(1, 70000))
        k = i % 3
        if k == 0: a = (np.uint64(1 << 40) + np.uint64(1000) * np.arange(n, dtype=np.uint64) + rng.integers(0, 512, n).astype(np.uint64))
        elif k == 1: a = rng.standard_normal(n).astype(np.float32)
        else: a = (rng.pareto(0.5, n) * 10).clip(0, 2e9).astype(np.int32)
        arrays.append(a)
    srcs = [torch.from_numpy(a.view(np.uint8)).cuda() for a in arrays]
    caps = [(L.pco_gfx_guarantee_chunk_size(a.size, G.DTYPE_BYTE[a.dtype.name]) + 64 + 15) // 16 * 16 for a in arrays]
    # explicit consecutive delta (16-bit latents hold for the ramps and fail for the rest: both split passes in one call), the
    # Auto specs (per-chunk deltas, lookback among them: chunks that never speculate next to chunks that do), lookback for all
    for kw in (dict(mode=1, delta=2, delta_order=1), dict(), dict(mode=1, delta=3)):
        gcfg = G.make_config(**kw)
        dsts = [torch.zeros(c, dtype=torch.uint8, device="cuda") for c in caps]
        tasks = (G.EncodeTask * len(arrays))(*[G.EncodeTask(s.data_ptr(), a.size, d.data_ptr(), c, G.DTYPE_BYTE[a.dtype.name], 0)
                                               for a, s, d, c in zip(arrays, srcs, dsts, caps)])
        res = (G.TaskResult * len(arrays))()
        G.check(L.pco_gfx_compress_chunks(len(arrays), tasks, C.byref(gcfg), res, None, None))
        outs = [torch.empty(a.nbytes, dtype=torch.uint8, device="cuda") for a in arrays]
        dtasks = (G.DecodeTask * len(arrays))(*[G.DecodeTask(d.data_ptr(), res[i].n_out, o.data_ptr(), a.size, G.DTYPE_BYTE[a.dtype.name], 0)
                                                for i, (a, d, o) in enumerate(zip(arrays, dsts, outs))])
        dres = (G.TaskResult * len(arrays))()
        G.check(L.pco_gfx_decompress_chunks(len(arrays), dtasks, dres, None, None))
        for i, a in enumerate(arrays):
            want = O.simple_compress(a, O.make_config(**kw))
            got = bytes(dsts[i][: res[i].n_out].cpu().numpy())
            hdr = len(want) - 1 - len(got)
            assert got == want[hdr:-1], (kw, i)
            assert dres[i].n_out == a.size and dres[i].consumed == res[i].n_out
            assert U.bits_equal(outs[i].cpu().numpy().view(a.dtype), a), (kw, i)
    # a .pco file assembled from device-produced chunks with the library's framing == the oracle's file
    a = arrays[0]
    hdr = np.zeros(32, np.uint8); k = L.pco_gfx_write_standalone_header(hdr.ctypes.data_as(C.c_void_p), 32, a.size, 0)
    assert bytes(hdr[:k]) + bytes(dsts[0][: res[0].n_out].cpu().numpy()) + b"\x00" == O.simple_compress(a, O.make_config(**kw))


def test_batched_api_asynchronous_form(L):
    """results == NULL + a device results array: nothing is read back mid-pipeline, so every kernel of
    the encode pipeline is launched (both split passes, the sort histogram with buffers allocated up front)."""
    import torch
    rng = np.random.default_rng(5)
    arrays = []
    for i in range(24):
        n = int(rng.integers(1, 50000))
        k = i % 4
        if k == 0: a = (np.uint64(1 << 40) + np.uint64(1000) * np.arange(n, dtype=np.uint64) + rng.integers(0, 512, n).astype(np.uint64))
        elif k == 1: a = rng.standard_normal(n).astype(np.float32)
        elif k == 2: a = (rng.integers(1000, 10000, n) / 100.0)
        else: a = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
        arrays.append(a)
    res_dt = np.dtype([("n_out", "<u8"), ("consumed", "<u8"), ("status", "<u4"), ("aux", "<u4")])
    srcs = [torch.from_numpy(a.view(np.uint8)).cuda() for a in arrays]
    caps = [(L.pco_gfx_guarantee_chunk_size(a.size, G.DTYPE_BYTE[a.dtype.name]) + 64 + 15) // 16 * 16 for a in arrays]
    for kw in (dict(mode=1, delta=2, delta_order=1), dict(mode=1, delta=1)):
        gcfg = G.make_config(**kw)
        dsts = [torch.zeros(c, dtype=torch.uint8, device="cuda") for c in caps]
        tasks = (G.EncodeTask * len(arrays))(*[G.EncodeTask(s.data_ptr(), a.size, d.data_ptr(), c, G.DTYPE_BYTE[a.dtype.name], 0)
                                               for a, s, d, c in zip(arrays, srcs, dsts, caps)])
        d_res = torch.zeros(len(arrays) * res_dt.itemsize, dtype=torch.uint8, device="cuda")
        G.check(L.pco_gfx_compress_chunks(len(arrays), tasks, C.byref(gcfg), None, C.c_void_p(d_res.data_ptr()), None))
        torch.cuda.synchronize()
        res = d_res.cpu().numpy().view(res_dt)
        assert (res["status"] == 0).all()
        outs = [torch.empty(a.nbytes, dtype=torch.uint8, device="cuda") for a in arrays]
        dtasks = (G.DecodeTask * len(arrays))(*[G.DecodeTask(d.data_ptr(), int(res["n_out"][i]), o.data_ptr(), a.size, G.DTYPE_BYTE[a.dtype.name], 0)
                                                for i, (a, d, o) in enumerate(zip(arrays, dsts, outs))])
        d_dres = torch.zeros(len(arrays) * res_dt.itemsize, dtype=torch.uint8, device="cuda")
        G.check(L.pco_gfx_decompress_chunks(len(arrays), dtasks, None, C.c_void_p(d_dres.data_ptr()), None))
        torch.cuda.synchronize()
        dres = d_dres.cpu().numpy().view(res_dt)
        for i, a in enumerate(arrays):
            want = O.simple_compress(a, O.make_config(**kw))
            got = bytes(dsts[i][: int(res["n_out"][i])].cpu().numpy())
            hdr = len(want) - 1 - len(got)
            assert got == want[hdr:-1], (kw, i)
            assert dres["status"][i] == 0 and dres["n_out"][i] == a.size
            assert U.bits_equal(outs[i].cpu().numpy().view(a.dtype), a), (kw, i)


def test_wrapped_surface_round_trip(L):
    """wrapped::ChunkCompressor / ChunkDecompressor (tests/low_level.rs:39-129): multi-page chunk, page by page."""
    rng = np.random.default_rng(8)
    nums = np.cumsum(rng.integers(-5, 50, 10000)).astype(np.int64)
    cfg = G.make_config(mode=1, delta=2, delta_order=2, max_page_n=3000)
    cc = C.c_void_p()
    G.check(L.pco_chunk_compressor_new(nums.ctypes.data_as(C.c_void_p), C.c_size_t(nums.size), C.c_ubyte(4), C.byref(cfg), C.byref(cc)))
    L.pco_chunk_compressor_n_pages.restype = C.c_size_t; L.pco_chunk_compressor_page_n.restype = C.c_size_t
    L.pco_chunk_compressor_meta_size_hint.restype = C.c_size_t; L.pco_chunk_compressor_page_size_hint.restype = C.c_size_t
    n_pages = L.pco_chunk_compressor_n_pages(cc)
    assert n_pages == 4
    page_ns = [L.pco_chunk_compressor_page_n(cc, C.c_size_t(i)) for i in range(n_pages)]
    assert page_ns == [2500] * 4 and sum(page_ns) == nums.size
    buf = np.zeros(1 << 20, np.uint8); w = C.c_size_t(0)
    G.check(L.pco_chunk_compressor_write_meta(cc, buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size), C.byref(w)))
    meta = bytes(buf[: w.value])
    assert len(meta) <= L.pco_chunk_compressor_meta_size_hint(cc)
    pages = []
    for i in range(n_pages):
        G.check(L.pco_chunk_compressor_write_page(cc, C.c_size_t(i), buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size), C.byref(w)))
        pages.append(bytes(buf[: w.value]))
        assert L.pco_chunk_compressor_page_size_hint(cc, C.c_size_t(i)) > 0
    L.pco_chunk_compressor_free(cc)
    # decode with the wrapped decompressor, page by page
    cd = C.c_void_p(); used = C.c_size_t(0)
    stream = meta + b"".join(pages)
    sbuf = np.frombuffer(stream, np.uint8)
    G.check(L.pco_chunk_decompressor_new(sbuf.ctypes.data_as(C.c_void_p), C.c_size_t(len(stream)), C.c_ubyte(4), C.c_uint8(4), C.byref(cd), C.byref(used)))
    assert used.value == len(meta)
    pos = len(meta); out = []
    for i in range(n_pages):
        dst = np.zeros(page_ns[i], np.int64); npr = C.c_size_t(0); cons = C.c_size_t(0)
        G.check(L.pco_chunk_decompressor_read_page(cd, C.c_void_p(sbuf.ctypes.data + pos), C.c_size_t(len(stream) - pos), C.c_size_t(page_ns[i]),
                                                   dst.ctypes.data_as(C.c_void_p), C.c_size_t(dst.size), C.byref(npr), C.byref(cons)))
        assert npr.value == page_ns[i] and cons.value == len(pages[i])
        pos += cons.value; out.append(dst)
    L.pco_chunk_decompressor_free(cd)
    assert np.array_equal(np.concatenate(out), nums)
    # a single-page wrapped chunk equals the standalone chunk minus its 4-byte preamble
    one = nums[:2000].copy()
    cfg1 = G.make_config(mode=1, delta=2, delta_order=2)
    G.check(L.pco_chunk_compressor_new(one.ctypes.data_as(C.c_void_p), C.c_size_t(one.size), C.c_ubyte(4), C.byref(cfg1), C.byref(cc)))
    G.check(L.pco_chunk_compressor_write_meta(cc, buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size), C.byref(w))); m1 = bytes(buf[: w.value])
    G.check(L.pco_chunk_compressor_write_page(cc, C.c_size_t(0), buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size), C.byref(w))); p1 = bytes(buf[: w.value])
    L.pco_chunk_compressor_free(cc)
    want = O.simple_compress(one, O.make_config(mode=1, delta=2, delta_order=2))
    assert want[-1 - len(m1) - len(p1):-1] == m1 + p1


def test_python_mirror_round_trip(L):
    import pcodec_amd as P
    rng = np.random.default_rng(12345)
    for dtype in ("f4", "f8", "i2", "i4", "i8", "u2", "u4", "u8"):  # pco_python/test/test_standalone.py:19-44
        data = rng.uniform(0, 1000, size=900).astype(dtype)
        cfg = P.ChunkConfig(mode_spec=P.ModeSpec.classic(), delta_spec=P.DeltaSpec.try_consecutive(1), paging_spec=P.PagingSpec.equal_pages_up_to(300))
        comp = P.standalone.simple_compress(data, cfg)
        np.testing.assert_array_equal(P.standalone.simple_decompress(comp), data)
        out = np.zeros(3, dtype)
        pr = P.standalone.simple_decompress_into(comp, out)
        assert pr.n_processed == 3 and not pr.finished
        np.testing.assert_array_equal(out, data[:3])
    with pytest.raises(RuntimeError, match="does not match chunk's number type"):
        P.standalone.simple_decompress_into(comp, np.zeros(10, np.float64))


def test_many_chunk_properties_full_size(L):
    """At BASELINE size (2^18 x many chunks) check size-independent properties: encode->decode round trip,
    determinism (same bytes for the same chunk wherever it sits in the batch), and chunk independence."""
    import torch
    nch = 64
    base = U.synth("c2")
    rng = np.random.default_rng(77)
    host = np.stack([base + np.uint64(rng.integers(0, 1 << 20)) for _ in range(nch)])
    host[17] = host[3]
    src = torch.from_numpy(host.view(np.int64)).cuda()
    dtb = 2
    cap = (L.pco_gfx_guarantee_chunk_size(base.size, dtb) + 64 + 15) // 16 * 16
    comp = torch.zeros(nch * cap, dtype=torch.uint8, device="cuda")
    tasks = (G.EncodeTask * nch)(*[G.EncodeTask(src.data_ptr() + i * base.nbytes, base.size, comp.data_ptr() + i * cap, cap, dtb, 0) for i in range(nch)])
    res = (G.TaskResult * nch)()
    gcfg, ocfg = U.cfg_pair("c2")
    G.check(L.pco_gfx_compress_chunks(nch, tasks, C.byref(gcfg), res, None, None))
    c3 = bytes(comp[3 * cap: 3 * cap + res[3].n_out].cpu().numpy()); c17 = bytes(comp[17 * cap: 17 * cap + res[17].n_out].cpu().numpy())
    assert c3 == c17
    want = O.simple_compress(host[3], ocfg)
    assert c3 == want[len(want) - 1 - len(c3):-1]
    out = torch.empty_like(src)
    dt = (G.DecodeTask * nch)(*[G.DecodeTask(comp.data_ptr() + i * cap, res[i].n_out, out.data_ptr() + i * base.nbytes, base.size, dtb, 0) for i in range(nch)])
    dres = (G.TaskResult * nch)()
    G.check(L.pco_gfx_decompress_chunks(nch, dt, dres, None, None))
    assert torch.equal(out, src)
    assert all(r.n_out <= L.pco_gfx_guarantee_chunk_size(base.size, dtb) for r in res)


def test_workspace_budget_cuts_a_call_into_sub_batches(L):
    """PCO_GFX_WORKSPACE_GB: a synchronous batched call whose scratch would exceed the budget runs as consecutive sub-batches through the
    same buffers -- same bytes, same results order."""
    import subprocess, sys
    code = r'''
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, gpu_util as U, oracle_lib as O
from pcodec_amd import _lib as G
rng = np.random.default_rng(9)
arrays = [(np.uint64(1 << 40) + np.uint64(1000) * np.arange(20000, dtype=np.uint64) + rng.integers(0, 512, 20000).astype(np.uint64)) for _ in range(300)]
kw = dict(mode=1, delta=2, delta_order=1)
chunks, back = U.gpu_batched(arrays, G.make_config(**kw))
for a, ch, b in zip(arrays, chunks, back):
    want = O.simple_compress(a, O.make_config(**kw))
    assert ch == U.chunk_of_file(want, len(ch)) and U.bits_equal(a, b)
print("ok")
'''
    env = dict(os.environ, PCO_GFX_WORKSPACE_GB="0.05")   # 50 MB: ~70 chunks of 20000 u64 per pass
    out = subprocess.run([sys.executable, "-c", code], cwd=os.path.join(HERE, ".."), env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_chunks_beyond_one_page_of_the_standalone_writer(L):
    """A batched task is one chunk of up to 2^24 numbers (the standalone writer would cut it at 2^18): 2^18 < n <= 2^22 against the
    oracle's single-chunk file, u64 / f32 / i16."""
    rng = np.random.default_rng(10)
    arrays = [(np.uint64(1 << 40) + np.uint64(1000) * np.arange(n, dtype=np.uint64) + rng.integers(0, 512, n).astype(np.uint64)) for n in ((1 << 18) + 7, 1 << 20)]
    arrays += [rng.standard_normal((1 << 19) + 1).astype(np.float32), (np.cumsum(rng.integers(-3, 4, 1 << 22)) % 30000).astype(np.int16)]
    for kw in (dict(mode=1, delta=2, delta_order=1), dict(mode=1, delta=1)):
        chunks, back = U.gpu_batched(arrays, G.make_config(**kw))
        for a, ch, b in zip(arrays, chunks, back):
            want = O.simple_compress(a, O.make_config(max_page_n=a.size, **kw))
            assert ch == U.chunk_of_file(want, len(ch)), (a.dtype, a.size, kw)
            assert U.bits_equal(a, b), (a.dtype, a.size, kw)


@pytest.mark.parametrize("level", [9, 10, 11, 12])
def test_levels_nine_to_twelve_full_size(L, level):
    """unoptimized_bins_log 9..11 at n = 2^18 (wrapped/chunk_compressor.rs:362-371: level 12 gives 11 there; 12 needs n >= 2^20, below):
    up to 2048 bins, tANS tables of up to 2^12 states -- every BASELINE config and a two-variable + lookback mix, byte-identical."""
    for kind in ("c2", "c3", "c1", "c4"):
        nums = U.synth(kind)
        kw = {"c1": dict(mode=1, delta=1), "c2": dict(mode=1, delta=2, delta_order=1), "c3": dict(mode=2, mode_f64=0.01, delta=1), "c4": dict(mode=1, delta=3)}[kind]
        want = O.simple_compress(nums, O.make_config(level=level, **kw))
        got = U.gpu_simple_compress(nums, G.make_config(level=level, **kw))
        assert got == want, (level, kind)
        assert U.bits_equal(U.gpu_simple_decompress(got, nums.dtype, nums.size), nums), (level, kind)
    rng = np.random.default_rng(level)
    for nums, kw in (((rng.integers(0, 1 << 20, 9000) * 8 + rng.integers(0, 3, 9000)).astype(np.uint32), dict(mode=4, mode_u64=8, delta=2, delta_order=1)),
                     (rng.standard_normal(70000).astype(np.float32), dict()),
                     (np.where(rng.random(100000) < 0.5, 7, rng.integers(0, 1 << 40, 100000)).astype(np.uint64), dict(mode=1, delta=1)),
                     (rng.integers(0, 1 << 62, (1 << 20) + 5, dtype=np.uint64), dict(mode=1, delta=1, max_page_n=1 << 21))):   # 4096 bins
        want = O.simple_compress(nums, O.make_config(level=level, **kw))
        got = U.gpu_simple_compress(nums, G.make_config(level=level, **kw))
        assert got == want, (level, nums.dtype, nums.size, kw)
        assert U.bits_equal(U.gpu_simple_decompress(got, nums.dtype, nums.size), nums), (level, nums.dtype, nums.size, kw)


def test_walk_arrangements_of_small_and_large_launches(L):
    """encode_fast.hip's two walk arrangements: up to 8192 (page, variable) items every walk goes through enc_walkd_kernel (narrow
    16-bit variables looked up by the block's gathering wave, the rest staged from enc_dissect_kernel's symbols); beyond that the
    variables with small tANS tables go 16 per wave through enc_walk_kernel<16>.  Both launch sizes, chunk lengths around the batch
    size (a page's last batch is partial; one-number chunks), narrow / wide / constant / two-variable chunks side by side: every chunk
    equals the oracle's bytes."""
    rng = np.random.default_rng(77)

    def make(i):
        n = int([1, 2, 255, 256, 257, 300, 511, 513, 1000, 1537][i % 10])
        kind = i % 7
        if kind == 0: return (np.uint64(1 << 40) + np.uint64(1000) * np.arange(n, dtype=np.uint64) + rng.integers(0, 512, n).astype(np.uint64))   # narrow after delta 1
        if kind == 1: return rng.integers(0, 1 << 62, n, dtype=np.uint64)                                                                      # wide
        if kind == 2: return np.full(n, 12345, np.uint64)                                                                                        # constant
        if kind == 3: return (rng.integers(0, 3000, n) + (1 << 20)).astype(np.uint64)                                                            # narrow, many bins
        if kind == 4: return (rng.integers(0, 40, n) * 7 + 3).astype(np.uint64)                                                                  # few values
        if kind == 5: return np.cumsum(rng.integers(-5, 6, n)).astype(np.int64).view(np.uint64)                                                  # random walk
        return (rng.integers(0, 5000, n) + (1 << 33)).astype(np.uint64)                                                                          # range just above 4096
    for count, kw in ((300, dict(mode=1, delta=2, delta_order=1)), (9100, dict(mode=1, delta=2, delta_order=1)), (9100, dict(mode=4, mode_u64=7, delta=1))):
        arrays = [make(i) for i in range(count)]
        chunks, back = U.gpu_batched(arrays, G.make_config(**kw))
        for i in range(count):
            assert U.bits_equal(back[i], arrays[i]), (count, kw, i)
        for i in list(range(0, count, max(1, count // 140))) + list(range(min(count, 70))):   # the oracle on a spread of them (every kind and length)
            want = O.simple_compress(arrays[i], O.make_config(**kw))
            assert chunks[i] == U.chunk_of_file(want, len(chunks[i])), (count, kw, i, arrays[i].size)


def test_one_variable_launch_beyond_8192_items_stays_with_the_fused_walk(L):
    """More than 8192 one-variable items whose value -> bin tables are allocated (chunks of 2 k+ numbers): every item stays with the fused kernels
    (enc_walkp_kernel's blocks where all sixteen items qualify, enc_walkd_kernel for the rest -- wide, constant and many-bin chunks break blocks
    up on purpose), enc_walk_kernel<16> is not launched; bytes equal the oracle's on a spread of chunks, all of them round-trip."""
    rng = np.random.default_rng(79)

    def make(i):
        n = int([2049, 2304, 2560, 2817, 3000, 4097][i % 6])
        kind = (i // 40) % 5 if (i // 200) % 2 else 0   # runs of 200 narrow chunks (whole walkp blocks), then 200 in runs of 40 of each kind
        if kind == 0: return (np.uint64(1 << 40) + np.uint64(1000) * np.arange(n, dtype=np.uint64) + rng.integers(0, 512, n).astype(np.uint64))
        if kind == 1: return rng.integers(0, 1 << 62, n, dtype=np.uint64)
        if kind == 2: return np.full(n, 12345, np.uint64)
        if kind == 3: return np.cumsum(rng.integers(0, 3000, n)).astype(np.uint64)
        return (np.uint64(7) * np.arange(n, dtype=np.uint64) + (rng.integers(0, 2, n) * 5000).astype(np.uint64))
    count, kw = 8300, dict(mode=1, delta=2, delta_order=1)
    arrays = [make(i) for i in range(count)]
    L.pco_gfx_profile_begin()
    chunks, back = U.gpu_batched(arrays, G.make_config(**kw))
    names = U.profile_names(L)
    assert "enc_walkp_kernel" in names and "enc_walk16_kernel" not in names, names
    for i in range(count):
        assert U.bits_equal(back[i], arrays[i]), i
    for i in list(range(0, count, 60)) + list(range(190, 260)):
        want = O.simple_compress(arrays[i], O.make_config(**kw))
        assert chunks[i] == U.chunk_of_file(want, len(chunks[i])), (i, arrays[i].size)


def test_many_short_chunks_keep_the_dissect_kernel(L):
    """The value -> bin tables of enc_walkd_kernel are 8 KB per (chunk, variable): a call whose chunks are much shorter than that does not
    allocate them (the symbols come from enc_dissect_kernel, the walk from the same kernel) -- same bytes either way."""
    rng = np.random.default_rng(78)
    arrays = [(np.uint64(1 << 30) + np.cumsum(rng.integers(0, 9, int(rng.integers(1, 700)))).astype(np.uint64)) for _ in range(7000)]   # 7000 x <= 5.6 KB against 56 MB of tables
    kw = dict(mode=1, delta=2, delta_order=1)
    chunks, back = U.gpu_batched(arrays, G.make_config(**kw))
    for i in range(len(arrays)):
        assert U.bits_equal(back[i], arrays[i]), i
    for i in range(0, len(arrays), 50):
        want = O.simple_compress(arrays[i], O.make_config(**kw))
        assert chunks[i] == U.chunk_of_file(want, len(chunks[i])), (i, arrays[i].size)


@pytest.mark.parametrize("kind", ["c2", "c3", "periodic8", "geometric32"])
def test_identical_chunks_in_one_call_give_identical_bytes(L, kind):
    """Hundreds of copies of one chunk in ONE call: every copy's bytes are the oracle's.  The copies sit in different walk blocks, at different
    LDS slots and on different CUs, and run at slightly different times -- a fault that depends on timing or placement (round 6 had one: a build
    of enc_walkp_kernel garbled a batch's tANS section in some blocks only) shows up as copies that differ from each other."""
    rng = np.random.default_rng(5)
    if kind in ("c2", "c3"):
        nums = U.synth(kind, 1 << 16); gcfg, ocfg = U.cfg_pair(kind)
    elif kind == "periodic8":
        base = rng.integers(0, 250, 37); n = 50000
        nums = ((base[np.arange(n) % 37] + rng.integers(0, 3, n)) % 256).astype(np.uint8)
        kw = dict(level=7, mode=1, delta=1, enable_8_bit=True); gcfg, ocfg = G.make_config(**kw), O.make_config(**kw)
    else:
        nums = (rng.geometric(0.02, 40000) % 100000).astype(np.uint32)
        kw = dict(mode=1, delta=1); gcfg, ocfg = G.make_config(enable_8_bit=True, **kw), O.make_config(**kw)
    k = 384
    chunks, back = U.gpu_batched([nums] * k, gcfg)
    ref = O.simple_compress(nums, ocfg)
    want = U.chunk_of_file(ref, len(chunks[0]))
    bad = [i for i, c in enumerate(chunks) if c != want]
    assert not bad, (kind, len(bad), bad[:8])
    assert all(U.bits_equal(b, nums) for b in back[:16])
