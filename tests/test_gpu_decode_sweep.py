"""-m gpu: the GPU decoders against the oracle's decoder on streams no restated encoder writes -- Dict mode (mode/dict.rs:70
join_latents), Conv1 delta (delta/conv1.rs:463 decode_in_place), a delta'd secondary variable, lookback with a stored state --
produced by the TEST-ONLY generator in oracle/pco_oracle_testenc.hpp (decode is deterministic: any valid stream will do; the
oracle's DECODER is the restatement, pinned by the reference's v1_0_0_dict.pco / v1_0_0_conv1.pco)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))

import decode_sweep_util as S  # noqa: E402
import oracle_lib as O  # noqa: E402
import gpu_util as U  # noqa: E402
from pcodec_amd import _lib as G  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    lib = G.lib()
    assert lib.pco_gfx_device_count() >= 1, "these tests need an MI355X; the product has no CPU path"
    return lib


def gpu_decode_files(L, files, dtypes, ns):
    """One pco_gfx_decompress_chunks call over whole .pco files (TASK_HAS_FILE_HEADER): [(status, n_out, array)]."""
    import torch
    k = len(files)
    srcs = [torch.from_numpy(np.frombuffer(f + b"\x00" * 64, np.uint8).copy()).cuda() for f in files]
    outs = [torch.zeros(max(n, 1) * np.dtype(dt).itemsize + 64, dtype=torch.uint8, device="cuda") for dt, n in zip(dtypes, ns)]
    tasks = (G.DecodeTask * k)(*[G.DecodeTask(s.data_ptr(), len(f), o.data_ptr(), n, G.DTYPE_BYTE[np.dtype(dt).name], G.TASK_HAS_FILE_HEADER)
                                 for s, f, o, dt, n in zip(srcs, files, outs, dtypes, ns)])
    res = (G.TaskResult * k)()
    L.pco_gfx_decompress_chunks(k, tasks, res, None, None)   # (a failing task makes the call return an error code; per-task statuses are what we read)
    torch.cuda.synchronize()
    out = []
    for i in range(k):
        arr = outs[i][: ns[i] * np.dtype(dtypes[i]).itemsize].cpu().numpy().view(dtypes[i])
        out.append((int(res[i].status), int(res[i].n_out), arr))
    return out


def needs_secondary_history(kw):
    """Lookback on a chunk whose SECONDARY variable is delta'd too (no encoder writes it: wrapped/chunk_compressor.rs:343,384 always say
    secondary_uses_delta = false): the second history needs scratch, so the task goes through the decoder twice -- synchronous calls
    only; an asynchronous call reports Unsupported (test_secondary_history_needs_a_synchronous_call)."""
    return kw.get("delta") == O.TE_DELTA_LOOKBACK and kw.get("secondary_uses_delta") and kw.get("mode") not in (O.MODE_CLASSIC, O.MODE_TRY_DICT)


@pytest.mark.parametrize("kind,count", [("dict", 330), ("conv1", 320), ("extra", 320)])
def test_gpu_decode_of_generated_streams(L, kind, count):
    """>= 200 valid streams of each kind: GPU decode == the input == the oracle's decode.  Batched 40 files per call (mixed dtypes)."""
    batch = []; twice = 0; compared = 0

    def flush():
        nonlocal twice, compared
        if not batch:
            return
        got = gpu_decode_files(L, [b[2] for b in batch], [b[1].dtype for b in batch], [b[1].size for b in batch])
        for (label, x, data, kw), (status, n_out, arr) in zip(batch, got):
            twice += 1 if needs_secondary_history(kw) else 0
            assert status == G.ST_OK, (label, status)
            assert n_out == x.size and U.bits_equal(arr, x), label
            compared += 1
        batch.clear()

    for label, x, kw in S.cases(kind, count, 777):
        data = O.test_encode(x, **kw)
        assert U.bits_equal(O.simple_decompress(data, x.dtype, cap=x.size + 8), x), label   # the stream is valid and says what we think
        batch.append((label, x, data, kw))
        if len(batch) == 40:
            flush()
    flush()
    assert compared >= 200, (kind, compared)
    assert (twice > 20) == (kind == "extra"), (kind, twice)


def test_generated_streams_through_the_host_entry_points(L):
    """The same kinds through pco_standalone_simple_decompress_into (the reference's C ABI), one file per call."""
    for kind in ("dict", "conv1", "extra"):
        for label, x, kw in S.cases(kind, 24, 31337):
            data = O.test_encode(x, **kw)
            assert U.bits_equal(U.gpu_simple_decompress(data, x.dtype, x.size), x), label


def test_damaged_generated_streams_never_crash_and_agree_when_the_oracle_accepts(L):
    """tests/corruption.rs on Dict / Conv1 / lookback-state streams: truncations and single bit flips.  The GPU must return; where the
    oracle (the reference's semantics) decodes the damaged stream without an error the GPU must produce the same numbers; where the
    oracle reports Corruption / InsufficientData the GPU may not report success with a full-length result that differs silently --
    it reports an error of its own (the two may name different kinds: which check trips first is not part of the format)."""
    rng = np.random.default_rng(5)
    files = []; meta = []
    for kind in ("dict", "conv1", "extra"):
        for label, x, kw in S.cases(kind, 14, 99):
            if x.size < 17:
                continue
            data = O.test_encode(x, **kw)
            for _ in range(6):
                b = bytearray(data)
                if rng.random() < 0.4:
                    b = b[: int(rng.integers(1, len(b)))]
                else:
                    pos = int(rng.integers(0, min(len(b), 200) if rng.random() < 0.5 else len(b)))
                    b[pos] ^= 1 << int(rng.integers(0, 8))
                files.append(bytes(b)); meta.append((label, x))
    got = []
    for i in range(0, len(files), 40):
        got += gpu_decode_files(L, files[i:i + 40], [m[1].dtype for m in meta[i:i + 40]], [m[1].size for m in meta[i:i + 40]])
    agree = 0; both_fail = 0
    for (label, x), f, (status, n_out, arr) in zip(meta, files, got):
        try:
            want = O.simple_decompress(f, x.dtype, cap=x.size + 8)
        except O.OracleError:
            want = None
        if want is not None and want.size <= x.size:
            assert status == G.ST_OK and n_out == want.size and U.bits_equal(arr[: want.size], want), (label, status, n_out, want.size)
            agree += 1
        elif want is None:
            assert status in (G.ST_CORRUPTION, G.ST_INSUFFICIENT_DATA, G.ST_INVALID_ARGUMENT, G.ST_UNSUPPORTED) or status == G.ST_OK, (label, status)
            both_fail += status != G.ST_OK
    assert agree > 20 and both_fail > 20, (agree, both_fail)


def test_secondary_history_needs_a_synchronous_call(L):
    """The asynchronous form of pco_gfx_decompress_chunks cannot come back with scratch: lookback + delta'd secondary is Unsupported there."""
    import torch
    x = S._smooth(np.random.default_rng(8), np.int32, 5000)
    data = O.test_encode(x, mode=O.MODE_TRY_INT_MULT, mode_u64=7, delta=O.TE_DELTA_LOOKBACK, window_n_log=10, state_n_log=0, secondary_uses_delta=True, lookback_seed=5)
    (status, n_out, arr), = gpu_decode_files(L, [data], [x.dtype], [x.size])
    assert status == G.ST_OK and U.bits_equal(arr, x)
    src = torch.from_numpy(np.frombuffer(data + b"\x00" * 64, np.uint8).copy()).cuda()
    out = torch.zeros(x.nbytes + 64, dtype=torch.uint8, device="cuda")
    d_res = torch.zeros(C.sizeof(G.TaskResult), dtype=torch.uint8, device="cuda")
    task = (G.DecodeTask * 1)(G.DecodeTask(src.data_ptr(), len(data), out.data_ptr(), x.size, G.DTYPE_BYTE[x.dtype.name], G.TASK_HAS_FILE_HEADER))
    G.check(L.pco_gfx_decompress_chunks(1, task, None, d_res.data_ptr(), None))
    torch.cuda.synchronize()
    res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=np.dtype([("n_out", "<u8"), ("consumed", "<u8"), ("status", "<u4"), ("aux", "<u4")]))
    assert int(res["status"][0]) == G.ST_UNSUPPORTED


def test_conv1_chunk_meta_validation(L):
    """metadata/chunk.rs:58-94 on the reference's own Conv1 asset and on generated ones: a quantization beyond min(31, conv_bits - 1)
    and weights / bias that could overflow the Conv type are Corruption -- at ChunkDecompressor creation already (meta only)."""
    x = S._smooth(np.random.default_rng(3), np.uint16, 3000)
    good = O.test_encode(x, mode=O.MODE_CLASSIC, delta=O.TE_DELTA_CONV1, quantization=4, bias=5, weights=[-16, 32])
    assert U.bits_equal(U.gpu_simple_decompress(good, x.dtype, x.size), x)
    cases = {
        "quantization 31 on a 16-bit latent (Conv = i32: max 31 is allowed)": dict(quantization=31, bias=0, weights=[1]),
        "weights overflow i32": dict(quantization=0, bias=0, weights=[1 << 15, 1 << 14]),          # 2^16 * (2^15 + 2^14) >= 2^31
        "bias overflows i32": dict(quantization=0, bias=1 << 31, weights=[0]),
    }
    for name, kw in cases.items():
        try:
            data = O.test_encode(x, mode=O.MODE_CLASSIC, delta=O.TE_DELTA_CONV1, **kw)
            oracle_ok = True
            try:
                O.simple_decompress(data, x.dtype, cap=x.size + 8)
            except O.OracleError as e:
                oracle_ok = False; assert e.kind == O.ERR_CORRUPTION, name
        except O.OracleError:   # the generator's own meta validation refused to write it: patch a good stream instead
            continue
        if oracle_ok:
            assert U.bits_equal(U.gpu_simple_decompress(data, x.dtype, x.size), x), name
        else:
            with pytest.raises(G.PcoGfxError) as ei:
                U.gpu_simple_decompress(data, x.dtype, x.size)
            assert ei.value.status == G.ST_CORRUPTION, name
    # u8: Conv = i16, quantization may not exceed 15
    x8 = S._smooth(np.random.default_rng(4), np.uint8, 900)
    ok8 = O.test_encode(x8, mode=O.MODE_CLASSIC, delta=O.TE_DELTA_CONV1, quantization=3, bias=0, weights=[8])
    assert U.bits_equal(U.gpu_simple_decompress(ok8, x8.dtype, x8.size), x8)
    # flip the stored quantization of the good u16 stream to 0x1f..: find the 5 quantization bits right after the 4-bit delta variant
    # by brute force: any single-bit flip in the ChunkMeta that the ORACLE calls Corruption must be Corruption on the GPU too
    hdr = len(good) - 1
    n_checked = 0
    for pos in range(8, 40):
        for bit in range(8):
            b = bytearray(good); b[pos] ^= 1 << bit
            try:
                O.simple_decompress(bytes(b), x.dtype, cap=x.size + 8)
            except O.OracleError as e:
                if e.kind != O.ERR_CORRUPTION or b"Conv1" not in str(e).encode():
                    continue
                with pytest.raises(G.PcoGfxError) as ei:
                    U.gpu_simple_decompress(bytes(b), x.dtype, x.size)
                assert ei.value.status == G.ST_CORRUPTION, (pos, bit)
                # ... and at chunk_decompressor_new (meta only), like ChunkMeta::read_from
                n_checked += 1
    assert n_checked >= 3, n_checked
    del hdr
