"""The reference's inline per-stage known-answer tests, replayed against the oracle.
Each test cites the reference test it restates (paths under /root/reference/pco/src)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O


def test_ans_spec_new():  # ans/spec.rs:96-108
    assert O.spread_state_symbols([1, 1, 3, 11]) == [0, 3, 2, 3, 2, 3, 3, 3, 3, 1, 3, 2, 3, 3, 3, 3]
    assert O.spread_state_symbols([1]) == [0]
    assert O.spread_state_symbols([2]) == [0, 0]


def test_quantize_weights_to():  # ans/encoding.rs:181-196
    assert O.quantize_weights_to([777], 777, 0)[:1] == [1]
    assert O.quantize_weights_to([777, 1], 778, 1) == [1, 1]
    assert O.quantize_weights_to([777, 1], 778, 2) == [3, 1]
    assert O.quantize_weights_to([2, 3, 6, 5, 1], 17, 3) == [1, 1, 3, 2, 1]
    assert O.quantize_weights_to([1, 1], 2, 1) == [1, 1]


def test_quantize_weights():  # ans/encoding.rs:199-206
    assert O.quantize_weights([77, 100], 177, 4) == (4, [7, 9])
    assert O.quantize_weights([77, 77], 154, 4) == (1, [1, 1])


def test_histogram_quicksort():  # histograms.rs:429-497 (shuffles via numpy; results are order independent)
    assert O.histogram(np.array([8], np.uint32), 0)[0] == [(1, 8, 8)]
    for seed in range(16):
        rng = np.random.default_rng(seed)
        lat = rng.permutation(100).astype(np.uint32)
        assert O.histogram(lat, 2)[0] == [(25, 0, 24), (25, 25, 49), (25, 50, 74), (25, 75, 99)]
        lat = np.zeros(100, np.uint32); lat[0] = 1; rng.shuffle(lat)
        assert O.histogram(lat, 2)[0] == [(99, 0, 0), (1, 1, 1)]
        lat = np.ones(100, np.uint32); lat[0] = 0; rng.shuffle(lat)
        assert O.histogram(lat, 2)[0] == [(1, 0, 0), (99, 1, 1)]
        lat = np.full(100, 5, np.uint32); lat[0] = 3; lat[1:3] = 7; rng.shuffle(lat)
        assert O.histogram(lat, 2)[0] == [(1, 3, 3), (97, 5, 5), (2, 7, 7)]
        assert O.histogram(lat, 1)[0] == [(98, 3, 5), (2, 7, 7)]
        lat = np.full(100, 5, np.uint32); lat[0:2] = 3; lat[2] = 7; rng.shuffle(lat)
        assert O.histogram(lat, 1)[0] == [(2, 3, 3), (98, 5, 7)]


@pytest.mark.parametrize("bits", [32, 64])
def test_histogram_multiset_rule_equals_literal_algorithm(bits):
    """The GPU implements the 'multiset rule'; it must equal the literal quickselect
    (histograms.rs:208-280) whenever the heapsort fallback did not run."""
    dt = O.NP_BITS_DTYPE[bits]
    rng = np.random.default_rng(1234 + bits)
    n_fallback = 0
    for trial in range(300):
        n = int(rng.integers(1, 3000))
        kind = trial % 6
        if kind == 0: lat = rng.integers(0, 1 << (bits - 1), n, dtype=np.uint64).astype(dt)
        elif kind == 1: lat = rng.integers(0, 8, n).astype(dt)
        elif kind == 2: lat = (rng.geometric(0.3, n) - 1).astype(dt)
        elif kind == 3: lat = np.where(rng.random(n) < 0.9, 7, rng.integers(0, 1000, n)).astype(dt)
        elif kind == 4: lat = np.sort(rng.integers(0, 50, n)).astype(dt)
        else: lat = (rng.integers(0, 20, n) * 1000 + rng.integers(0, 3, n)).astype(dt)
        for log in (0, 1, 4, 8):
            lit, fb = O.histogram(lat, log, rule=0)
            rule, _ = O.histogram(lat, log, rule=1)
            if fb:
                n_fallback += 1
                continue
            assert lit == rule, (trial, n, log)
            assert sum(b[0] for b in lit) == n
    assert n_fallback < 60


def test_bin_optimization():  # bin_optimization.rs:215-273
    bins = [(100, 1, 16), (100, 33, 48), (100, 49, 64), (100, 65, 74), (50, 75, 79)]
    assert O.optimize_bins(bins, 32, 10) == [(100, 1, 16, 4), (200, 33, 64, 5), (150, 65, 79, 4)]
    bins = [(1000, 0, 150), (1000, 200, 200)]
    assert O.optimize_bins(bins, 32, 10) == [(1000, 0, 150, 8), (1000, 200, 200, 0)]


def test_log2_approx():  # bin_optimization.rs:275-316
    lib = O.lib()
    for e in range(32):
        assert lib.pco_oracle_log2_approx(float(1 << e)) == float(e)
    prev = -np.inf
    for i in range(1, 101):
        v = lib.pco_oracle_log2_approx(float(i))
        assert v >= prev and abs(np.log2(np.float32(i)) - v) < 0.0076
        prev = v


def test_lookback_encode_kat():  # delta/lookback.rs:254-300
    lat = np.full(100, 100, np.uint32)
    lat[1] = 200; lat[2] = 201; lat[3] = 202; lat[5] = 203; lat[15] = 204; lat[50] = 205
    lb = O.choose_lookbacks(lat, window_n_log=4, state_n_log=1)
    assert len(lb) == 98
    assert lb[0] == 1 and lb[2] == 4 and lb[13] == 10 and lb[48] == 1
    deltas = lat.copy(); state = np.zeros(2, np.uint32)
    rc = O.lib().pco_oracle_lookback_encode_u32(deltas.ctypes.data_as(C.c_void_p), C.c_size_t(100), C.c_uint32(1),
                                               lb.ctypes.data_as(C.c_void_p), state.ctypes.data_as(C.c_void_p))
    assert rc == 0 and state.tolist() == [100, 200]


def test_consecutive_encode_kat():  # delta/consecutive.rs:57-78
    lat = np.array([2, 2, 1, 0xFFFFFFFF, 0], np.uint32); moments = np.zeros(2, np.uint32)
    rc = O.lib().pco_oracle_consecutive_encode_u32(lat.ctypes.data_as(C.c_void_p), C.c_size_t(5), C.c_size_t(2),
                                                  moments.ctypes.data_as(C.c_void_p))
    assert rc == 0
    assert moments.tolist() == [2, 0]
    # second differences of [2,2,1,MAX,0] = [-1, -1, 3] (wrapping), toggled by MID
    exp = (np.array([-1, -1, 3], np.int64) + (1 << 31)) % (1 << 32)
    assert lat[2:].tolist() == exp.tolist()


def test_choose_mode_sample_kat():  # sampling.rs:186-202 -- pins the xoroshiro128++ restatement
    nums = -np.arange(150, dtype=np.float32)
    idx = O.mode_sample_indices(150)
    sample = np.sort(nums[idx][nums[idx] != 0.0])
    assert len(sample) == 13
    assert sample[:3].tolist() == [-135.0, -131.0, -114.0]


def _cand(sample):
    s = np.array(sample, np.uint32); base = C.c_uint32(0); score = C.c_double(0); found = C.c_int(0)
    assert O.lib().pco_oracle_choose_candidate_base_u32(s.ctypes.data_as(C.c_void_p), C.c_size_t(len(s)), C.byref(base),
                                                        C.byref(score), C.byref(found)) == 0
    return base.value if found.value else None


def test_calc_candidate_gcd():  # mode/int_mult.rs:294-331
    assert _cand([0, 4, 8]) is None
    assert _cand([0, 4, 8, 10, 14, 18, 20, 24, 28]) == 4
    assert _cand([1, 11, 21, 31, 41, 51, 61, 71, 82]) == 10
    assert _cand([1, 11, 22, 31, 41, 51, 61, 71, 82]) is None
    rng = np.random.default_rng(0)
    assert _cand((rng.integers(0, 1000, 200) * 2).tolist()) == 2


def _quant_bid(sample):
    s = np.array(sample, np.float32); k = C.c_uint32(0); bs = C.c_double(0); found = C.c_int(0)
    assert O.lib().pco_oracle_float_quant_bid_f32(s.ctypes.data_as(C.c_void_p), C.c_size_t(len(s)), C.byref(k),
                                                  C.byref(bs), C.byref(found)) == 0
    return (k.value, bs.value) if found.value else None


def test_float_quant_compute_bid():  # mode/float_quant.rs:265-284
    assert _quant_bid(np.arange(100, dtype=np.float32)) == (17, 17.0)
    s = np.arange(100, dtype=np.float32); s[0] += np.float32(0.1); s[37] -= np.float32(0.1)
    k, bs = _quant_bid(s)
    assert k == 17 and 15.0 < bs < 17.0
    assert _quant_bid([0.0, 1.0] * 50) is None


def test_float_quant_split_specific_values():  # mode/float_quant.rs:176-224
    eps = np.finfo(np.float32).eps
    nums = np.array([-np.inf, -1.0 - eps, -1.0, -0.0, 0.0, 1.0, 1.0 + eps, np.inf], np.float32)
    prim, sec, mk, mp = O.split_latents(nums, O.make_config(mode=O.MODE_TRY_FLOAT_QUANT, mode_u64=5))
    assert mk == 3 and mp == 5
    assert prim.tolist() == [0b00000000000000111111111111111111, 0b00000010000000111111111111111111,
                             0b00000010000000111111111111111111, 0b00000011111111111111111111111111,
                             0b00000100000000000000000000000000, 0b00000101111111000000000000000000,
                             0b00000101111111000000000000000000, 0b00000111111111000000000000000000]
    assert sec.tolist() == [0, 1, 0, 0, 0, 0, 1, 0]
    # float_quant.rs:226-239
    nums = np.array([-2.345, -1.234, -0.0, 0.0, 1.234, 2.345], np.float32).astype(np.float64)
    _, sec, _, _ = O.split_latents(nums, O.make_config(mode=O.MODE_TRY_FLOAT_QUANT, mode_u64=53 - 24))
    assert not sec.any()


def test_int_mult_split():  # mode/int_mult.rs:255-277
    prim, sec, mk, mp = O.split_latents(np.array([8, 1, 5], np.uint32), O.make_config(mode=O.MODE_TRY_INT_MULT, mode_u64=4))
    assert (mk, mp) == (1, 4) and prim.tolist() == [2, 0, 1] and sec.tolist() == [0, 1, 1]


def test_auto_mode_choices():
    # data_types/float.rs:458-466
    nums = np.arange(2000, dtype=np.float64) * 1.5
    info, _, _ = O.chunk_plan(nums, O.make_config(delta=O.DELTA_NOOP))
    assert info.mode_kind == 2
    assert np.array([info.mode_base_latent], np.uint64)[0] == (np.array([1.5]).view(np.uint64)[0] ^ np.uint64(1 << 63))
    # data_types/float.rs:508-518
    lowest = np.array([1.0]).view(np.uint64)[0]
    nums = (lowest + (np.arange(1000, dtype=np.uint64) << np.uint64(20))).view(np.float64)
    info, _, _ = O.chunk_plan(nums, O.make_config(delta=O.DELTA_NOOP))
    assert (info.mode_kind, info.mode_k) == (3, 20)
    # tests/recovery.rs:389-401
    nums = np.arange(100, dtype=np.float32); nums[77] += np.float32(0.0001)
    info, _, _ = O.chunk_plan(nums, O.make_config())
    assert info.mode_kind == 2 and info.mode_base_latent == (0x3F800000 ^ 0x80000000)
    # tests/recovery.rs:376-386 (explicit float mult base)
    nums = np.array([100.1, 299.9, 200.0] * 100, np.float64)
    info, _, _ = O.chunk_plan(nums, O.make_config(mode=O.MODE_TRY_FLOAT_MULT, mode_f64=100.0))
    assert info.mode_kind == 2
    assert np.array([info.mode_base_latent], np.uint64)[0] == (np.array([100.0]).view(np.uint64)[0] ^ np.uint64(1 << 63))


def test_stability_bin_counts():  # tests/stability.rs:37-105
    cfg = O.make_config(mode=O.MODE_CLASSIC, delta=O.DELTA_NOOP)
    info, bins, _ = O.chunk_plan(np.array([0] * 50 + [1000] * 50, np.uint32), cfg)
    assert not info.var_present[0] and not info.var_present[2] and len(bins[1]) == 2
    info, bins, _ = O.chunk_plan(np.array([0] + [1] * ((1 << 16) + 1), np.uint32), cfg)
    assert len(bins[1]) == 2
    n = 1000
    nums = (np.uint64(0xFFFFFFFFFFFFFFFF // n) * np.arange(n, dtype=np.uint64))
    info, bins, _ = O.chunk_plan(nums, cfg)
    assert len(bins[1]) == 1 and bins[1][0][2] == 64
