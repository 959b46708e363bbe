"""Streams for the decode sweeps of Dict mode, Conv1 delta and the other format features no restated encoder writes
(oracle/pco_oracle_testenc.hpp, TEST-ONLY generator).  cases(kind, count, seed) yields (label, array, test_encode kwargs)."""
import numpy as np

import oracle_lib as O

ALL_DTYPES = [np.uint8, np.int8, np.uint16, np.int16, np.float16, np.uint32, np.int32, np.float32, np.uint64, np.int64, np.float64]
NARROW = [np.uint8, np.int8, np.uint16, np.int16, np.float16, np.uint32, np.int32, np.float32]   # Conv1: latents of at most 32 bits


def _sizes(rng):
    return int(rng.choice([1, 2, 3, 5, 17, 255, 256, 257, 511, 512, 513, 1000, 2049, 4097, 9000, 20011]))


def _chunks(rng, n):
    if n < 4 or rng.random() < 0.6:
        return [n]
    k = int(rng.integers(2, 4)); cuts = sorted(rng.choice(np.arange(1, n), size=k - 1, replace=False).tolist())
    return [b - a for a, b in zip([0] + cuts, cuts + [n])]


def _values(rng, dt, n, distinct):
    """n numbers of dtype dt drawn from `distinct` distinct values (runs, zipf-ish popularity)."""
    dt = np.dtype(dt)
    if dt.kind == "f":
        pool = rng.normal(size=distinct).astype(dt)
        pool[: min(3, distinct)] = np.array([0.0, -0.0, np.inf], dt)[: min(3, distinct)]
    else:
        info = np.iinfo(dt)
        pool = rng.integers(info.min, info.max, size=distinct, dtype=np.int64 if dt.kind == "i" else np.uint64, endpoint=True).astype(dt)
    pop = rng.zipf(1.5, size=n) % distinct
    if rng.random() < 0.3:   # runs
        pop = np.repeat(pop[: max(1, n // 7)], 7)[:n] if n >= 7 else pop
        pop = np.resize(pop, n)
    return pool[pop]


def _cap_state(kw):
    """A page shorter than the lookback state cannot be written (delta/lookback.rs:166-185 pads the state at the FRONT, and the
    decoder hands the first n state slots back): keep 2^state_n_log <= the shortest chunk."""
    if kw.get("delta") == O.TE_DELTA_LOOKBACK:
        kw["state_n_log"] = min(kw.get("state_n_log", 0), int(min(kw["chunks"])).bit_length() - 1)
    return kw


def _delta_kw(rng, n, allow_conv, latent_bits):
    r = rng.random()
    if r < 0.3:
        return {}
    if r < 0.5:
        return dict(delta=O.TE_DELTA_CONSECUTIVE, order=int(rng.integers(1, 4)))
    if r < 0.8 or not allow_conv:
        w = int(rng.integers(4, 16)); s = int(rng.integers(0, 3)) if rng.random() < 0.4 else 0
        return dict(delta=O.TE_DELTA_LOOKBACK, window_n_log=w, state_n_log=min(s, w), lookback_seed=int(rng.integers(1, 1 << 30)) if rng.random() < 0.7 else 0)
    return conv_kw(rng, latent_bits)


def conv_kw(rng, latent_bits):
    """Conv1 parameters inside metadata/chunk.rs:58-94's bounds: quantization <= min(31, conv_bits - 1),
    |bias| + 2^l_bits * sum|w| < 2^(conv_bits - 1)."""
    conv_bits = 64 if latent_bits == 32 else 2 * latent_bits
    order = int(rng.choice([1, 2, 2, 3, 4, 6, 6, 9]))
    room = (1 << (conv_bits - 1 - latent_bits)) - 1   # bound on sum |w| (bias takes a little of it)
    qmax = max(0, min(31, conv_bits - 1, (room // max(order, 1)).bit_length() - 2))
    q = int(rng.integers(0, qmax + 1))
    style = rng.random()
    if style < 0.4 and order >= 2 and 3 * (1 << q) < room:   # linear extrapolation: 2 x[i-1] - x[i-2]
        w = [0] * order; w[-1] = 2 << q; w[-2] = -(1 << q)
    elif style < 0.6 and (1 << q) < room:                    # previous value
        w = [0] * order; w[-1] = 1 << q
    else:
        per = max(1, min(room // (order + 1), (1 << 31) - 1))
        w = [int(x) for x in rng.integers(-per, per + 1, order)]
    used = sum(abs(x) for x in w)
    bias_room = ((1 << (conv_bits - 1)) - 1 - (used << latent_bits))
    bias_room = max(0, min(bias_room - 1, (1 << 62)))
    bias = int(rng.integers(-bias_room, bias_room + 1)) if bias_room > 0 and rng.random() < 0.7 else 0
    return dict(delta=O.TE_DELTA_CONV1, quantization=q, bias=bias, weights=w)


def dict_cases(count, seed):
    rng = np.random.default_rng(seed)
    for i in range(count):
        dt = ALL_DTYPES[i % len(ALL_DTYPES)]
        n = _sizes(rng)
        bits = np.dtype(dt).itemsize * 8
        distinct = int(min(rng.choice([1, 2, 3, 17, 200, 3000]), 1 << min(bits, 16)))
        x = _values(rng, dt, n, distinct)
        kw = dict(mode=O.MODE_TRY_DICT, chunks=_chunks(rng, n), dict_first_appearance=bool(rng.random() < 0.5), level=int(rng.choice([0, 4, 8, 12])))
        kw.update(_delta_kw(rng, n, True, 32))   # the dictionary indices are u32 latents (metadata/mode.rs:197-202)
        yield f"dict[{i}] {np.dtype(dt).name} n={n} distinct={distinct} {kw}", x, _cap_state(kw)


def _smooth(rng, dt, n):
    dt = np.dtype(dt)
    t = np.arange(n)
    y = 1000 * np.sin(t / 37.0) + 0.02 * t * t / max(n, 1) + rng.normal(scale=2.0, size=n)
    if dt.kind == "f":
        return y.astype(dt)
    info = np.iinfo(dt)
    scale = min(1.0, (info.max - info.min) / 4000.0)
    mid = (info.max + info.min) // 2
    return np.clip(np.rint(y * scale) + mid, info.min, info.max).astype(dt)


def conv_cases(count, seed):
    rng = np.random.default_rng(seed)
    for i in range(count):
        dt = np.dtype(NARROW[i % len(NARROW)])
        n = _sizes(rng)
        bits = dt.itemsize * 8
        x = _smooth(rng, dt, n) if rng.random() < 0.7 else _values(rng, dt, n, 50)
        kw = dict(chunks=_chunks(rng, n), level=int(rng.choice([0, 8])))
        r = rng.random()
        if dt.kind == "f" and r < 0.3 and bits >= 16:
            kw.update(mode=O.MODE_TRY_FLOAT_MULT, mode_f64=0.5)
        elif dt.kind == "f" and r < 0.5:
            kw.update(mode=O.MODE_TRY_FLOAT_QUANT, mode_u64=int(rng.integers(1, 10 if bits == 16 else 20)))
        elif dt.kind != "f" and r < 0.3:
            kw.update(mode=O.MODE_TRY_INT_MULT, mode_u64=int(rng.integers(2, 50)))
        else:
            kw.update(mode=O.MODE_CLASSIC)
        kw.update(conv_kw(rng, bits))
        yield f"conv1[{i}] {dt.name} n={n} {kw}", x, kw


def extra_cases(count, seed):
    """What the format allows and no encoder writes: a delta'd secondary variable (Consecutive and Lookback), lookback state_n_log > 0."""
    rng = np.random.default_rng(seed)
    for i in range(count):
        dt = np.dtype([np.uint16, np.int32, np.uint32, np.float32, np.int64, np.float64, np.uint64, np.float16][i % 8])
        n = _sizes(rng)
        x = _smooth(rng, dt, n)
        kw = dict(chunks=_chunks(rng, n))
        if rng.random() < 0.25:   # classic mode, lookback with a stored state of more than one latent
            w = int(rng.integers(4, 16))
            kw.update(mode=O.MODE_CLASSIC, delta=O.TE_DELTA_LOOKBACK, window_n_log=w, state_n_log=int(rng.integers(1, min(w, 5) + 1)),
                      lookback_seed=int(rng.integers(0, 1 << 30)) if rng.random() < 0.7 else 0)
            yield f"extra[{i}] {dt.name} n={n} {kw}", x, _cap_state(kw)
            continue
        if dt.kind == "f":
            if rng.random() < 0.5: kw.update(mode=O.MODE_TRY_FLOAT_MULT, mode_f64=0.25)
            else: kw.update(mode=O.MODE_TRY_FLOAT_QUANT, mode_u64=int(rng.integers(1, 9)))
        else:
            kw.update(mode=O.MODE_TRY_INT_MULT, mode_u64=int(rng.integers(2, 90)))
        if rng.random() < 0.5:
            kw.update(delta=O.TE_DELTA_CONSECUTIVE, order=int(rng.integers(1, 8)), secondary_uses_delta=True)
        else:
            w = int(rng.integers(4, 16))
            kw.update(delta=O.TE_DELTA_LOOKBACK, window_n_log=w, state_n_log=int(rng.integers(0, min(w, 4) + 1)), secondary_uses_delta=bool(rng.random() < 0.6),
                      lookback_seed=int(rng.integers(0, 1 << 30)) if rng.random() < 0.7 else 0)
        yield f"extra[{i}] {dt.name} n={n} {kw}", x, _cap_state(kw)


def cases(kind, count, seed):
    return {"dict": dict_cases, "conv1": conv_cases, "extra": extra_cases}[kind](count, seed)
