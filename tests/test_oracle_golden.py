"""Oracle vs the reference's golden .pco assets (decode pins + encode pins).

Expected arrays restate the deterministic generators of
/root/reference/pco/src/tests/compatibility.rs:71-303.
"""
import os

import numpy as np
import pytest

import oracle_lib as O

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_assets")


def asset(name):
    with open(os.path.join(ASSETS, name), "rb") as f:
        return f.read()


def bits_equal(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    u = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(u), b.view(u))


def pseudorandom_f16s():
    # compatibility.rs:129-143 (f32 arithmetic, then f16::from_f32 = round to nearest even)
    num = np.float32(0.1); out = []
    for _ in range(2000):
        num = np.float32(np.fmod(np.float32(np.float32(num * np.float32(77.7)) + np.float32(0.1)), np.float32(2.0)))
        out.append(np.float32(-1.0) - num if num < np.float32(1.0) else num)
    return np.array(out, dtype=np.float32).astype(np.float16)


def expected_arrays():
    e = {}
    e["v0_0_0_classic.pco"] = np.concatenate([np.arange(0, 1000), np.arange(2000, 3000)]).astype(np.int32)
    x = np.arange(2000, dtype=np.float32); x[1337] += np.float32(1.001)
    e["v0_0_0_delta_float_mult.pco"] = x
    x = (np.arange(2000) * 1000).astype(np.int32); x[1337] -= 1
    e["v0_1_0_delta_int_mult.pco"] = x
    e["v0_1_1_standalone_versioned.pco"] = np.zeros(0, np.float32)
    h = pseudorandom_f16s()
    e["v0_3_0_f16.pco"] = h
    f = h.astype(np.float32)
    bump = np.abs(f) < np.float32(1.1)
    fb = f.view(np.uint32).copy(); fb[bump] += 1
    e["v0_3_0_float_quant.pco"] = fb.view(np.float32)
    e["v0_4_0_lookback_delta.pco"] = np.tile(np.array(
        [1121827092, 729032807, 3968137854, 2875434067, 3775328080, 431649926, 1048116090, 1906978350, 14752788,
         1180462487], dtype=np.uint32), 100)
    e["v0_4_5_uniform_type.pco"] = np.array([1, 2, 3, 4, 5], np.uint32)
    e["v0_4_8_minor_version.pco"] = np.array([1, 2, 3, 4, 5], np.uint32)
    e["v1_0_0_u8.pco"] = np.concatenate([np.arange(0, 65), np.arange(192, 256)]).astype(np.uint8)
    e["v1_0_0_i8.pco"] = np.concatenate([np.arange(-128, -63), np.arange(64, 128)]).astype(np.int8)
    return e


EXPECTED = expected_arrays()


@pytest.mark.parametrize("name", sorted(EXPECTED))
def test_oracle_decodes_reference_asset(name):
    exp = EXPECTED[name]
    got = O.simple_decompress(asset(name), exp.dtype)
    assert bits_equal(got, exp), name


def expected_oracle_only():
    """Dict mode and Conv1 delta are outside the hot-path scope: the GPU product refuses them, but
    the oracle decodes them, so that every one of the reference's 13 golden assets pins it."""
    e = {}
    e["v1_0_0_dict.pco"] = np.array([8924659283, 234897984367, 9827358920] * 1000, dtype=np.uint64)   # compatibility.rs:248-259
    xm1 = 0.0; xm2 = 0.0; nums = []                                                                     # compatibility.rs:262-278 (f32 arithmetic)
    for i in range(2000):
        x = np.float32(np.float32(np.float32(np.float32(xm1) * np.float32(1.99)) - np.float32(xm2)) + np.float32((i * 47) % 77 - 38))
        nums.append(int(np.int32(np.float32(x + np.float32(10000.0)))))
        xm2 = xm1; xm1 = float(x)
    e["v1_0_0_conv1.pco"] = np.array(nums, dtype=np.int32)
    return e


EXPECTED_ORACLE_ONLY = expected_oracle_only()


@pytest.mark.parametrize("name", sorted(EXPECTED_ORACLE_ONLY))
def test_oracle_decodes_out_of_scope_assets(name):
    exp = EXPECTED_ORACLE_ONLY[name]
    assert bits_equal(O.simple_decompress(asset(name), exp.dtype), exp), name


def test_every_reference_asset_is_pinned():
    assert sorted(os.listdir(ASSETS)) == sorted(list(EXPECTED) + list(EXPECTED_ORACLE_ONLY))


def test_oracle_reports_corrupt_dict_and_conv1_streams():
    blob = bytearray(asset("v1_0_0_dict.pco"))
    for cut in (10, 20, 30, len(blob) - 3):
        with pytest.raises(O.OracleError):
            O.simple_decompress(bytes(blob[:cut]), np.uint64)
    blob = bytearray(asset("v1_0_0_conv1.pco"))
    for cut in (12, 40, len(blob) // 2):
        with pytest.raises(O.OracleError):
            O.simple_decompress(bytes(blob[:cut]), np.int32)


@pytest.mark.parametrize("name", ["v1_0_0_u8.pco", "v1_0_0_i8.pco"])
def test_oracle_reencodes_v1_assets_byte_for_byte(name):
    # written by simple_compress(nums, ChunkConfig::default().with_enable_8_bit(true)),
    # compatibility.rs:281-303: Auto mode, Auto delta, level 8, no uniform type
    enc = O.simple_compress(EXPECTED[name], O.make_config(), uniform_type=False)
    assert enc == asset(name)


def test_oracle_reencodes_older_assets_chunk_bytes():
    # Older standalone/format headers (8 bytes: magic, standalone v2, varint n_hint, format
    # version) vs ours (10 bytes: + uniform-type byte, + minor version), but the chunk encoding is
    # unchanged since: the bytes from the chunk's dtype byte onward must be identical.
    for name, cfg, hdr_old, hdr_new in [
        ("v0_4_0_lookback_delta.pco", O.make_config(delta=O.DELTA_TRY_LOOKBACK), 8, 10),
        ("v0_4_8_minor_version.pco", O.make_config(), 8, 10),
    ]:
        gold = asset(name)
        enc = O.simple_compress(EXPECTED[name], cfg)
        assert enc[hdr_new:] == gold[hdr_old:], name


def test_asset_metadata_matches_survey():
    info, bins = O.inspect_first_chunk(asset("v1_0_0_u8.pco"))
    assert (info.standalone_version, info.fmt_major, info.fmt_minor, info.n_hint, info.n) == (3, 4, 1, 129, 129)
    assert info.mode_kind == 0 and info.delta_kind == 1 and info.delta_order == 1
    assert info.ans_size_log[1] == 7
    assert bins[1].tolist() == [[1, 0, 0], [127, 129, 0]]
    assert info.meta_end_byte == 23
    info, bins = O.inspect_first_chunk(asset("v0_4_0_lookback_delta.pco"))
    assert info.delta_kind == 2 and info.window_n_log == 10 and info.state_n_log == 0
    assert info.ans_size_log[0] == 7 and bins[0].tolist() == [[1, 1, 4], [127, 10, 0]]
    assert info.ans_size_log[1] == 8 and len(bins[1]) == 3


def test_adversarial_order_sends_the_oracle_into_the_heapsort_branch():
    """histograms.rs:248-258: after 1 + floor(log2(n + 1)) bad pivots on one recursion path the reference heapsorts and switches to
    apply_sorted, whose tie rule differs from the quickselect path's.  No natural input reaches the branch (0 of 68 200 chunks in the
    census); tests/golden/hist_fallback.npz holds two orders built against the literal algorithm by a McIlroy-style adversary
    (scripts/make_hist_fallback_fixture.py) that do: the oracle reports the branch on them, its bins differ there from the multiset rule's
    (the quickselect path's result as a function of the sorted numbers -- what the GPU computes), and both encodings decode to the input."""
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    fx = np.load(os.path.join(here, "golden", "hist_fallback.npz"))
    spec = importlib.util.spec_from_file_location("mkfix", os.path.join(here, "..", "scripts", "make_hist_fallback_fixture.py"))
    mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
    differs = lambda x: O.histogram(x.copy(), 8, rule=0)[0] != O.histogram(x.copy(), 8, rule=1)[0]
    again, _, decided = mk.build(5000, 4, differs)          # the construction is deterministic: the committed array is its output
    assert np.array_equal(again, fx["n5000"]) and decided == 66
    cfg = O.make_config(mode=O.MODE_CLASSIC, delta=O.DELTA_NOOP)
    for key in ("n5000", "n262144"):
        x = fx[key]
        lit, fb = O.histogram(x.copy(), 8, rule=0)
        mul, _ = O.histogram(x.copy(), 8, rule=1)
        assert fb and lit != mul, key
        assert O.histogram(np.sort(x), 8, rule=0) == (mul, False)   # the same numbers in sorted order: no bad pivot, and the literal algorithm IS the multiset rule
        _, _, plan_fb = O.chunk_plan(x, cfg)
        assert plan_fb
        literal = O.simple_compress(x, cfg)
        O.set_hist_rule(1)
        try:
            multiset = O.simple_compress(x, cfg)
        finally:
            O.set_hist_rule(0)
        assert literal != multiset
        assert np.array_equal(O.simple_decompress(literal, np.uint32), x) and np.array_equal(O.simple_decompress(multiset, np.uint32), x)
