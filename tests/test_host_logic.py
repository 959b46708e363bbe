"""-m "not gpu": the product's HOST-side arithmetic that never touches the device, compiled on its own with g++ and checked on the CPU:
the binary16 type behind explicit f16 float-mult bases and f16 Auto (pcodec_amd/csrc/pco_half.h) against numpy's float16, and the
host path of Auto mode detection (pco_auto_host.inc: f16, oversized samples, overflowing GCD lists) against the oracle's bids."""
import ctypes as C
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "pcodec_amd", "csrc")
pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")


def _build(tmp_path, name, source, extra=()):
    src = tmp_path / (name + ".cpp"); src.write_text(source)
    out = tmp_path / ("lib" + name + ".so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-w", "-o", str(out), str(src), *extra])
    return C.CDLL(str(out))


def test_host_binary16_matches_numpy(tmp_path):
    lib = _build(tmp_path, "half", f'''
#include "{CSRC}/pco_half.h"
using namespace pcogfx;
extern "C" {{
  uint16_t h_from_f32(float f) {{ return F16::from_f32(f).bits; }}
  uint16_t h_from_f64(double d) {{ return F16::from_f64(d).bits; }}
  float h_to_f32(uint16_t b) {{ return F16::raw(b).to_f32(); }}
  uint16_t h_mul(uint16_t a, uint16_t b) {{ return (F16::raw(a) * F16::raw(b)).bits; }}
  uint16_t h_div(uint16_t a, uint16_t b) {{ return (F16::raw(a) / F16::raw(b)).bits; }}
}}''')
    lib.h_from_f32.restype = C.c_uint16; lib.h_from_f32.argtypes = [C.c_float]
    lib.h_from_f64.restype = C.c_uint16; lib.h_from_f64.argtypes = [C.c_double]
    lib.h_to_f32.restype = C.c_float; lib.h_to_f32.argtypes = [C.c_uint16]
    for f in (lib.h_mul, lib.h_div): f.restype = C.c_uint16; f.argtypes = [C.c_uint16, C.c_uint16]
    every = np.arange(65536, dtype=np.uint16)
    with np.errstate(all="ignore"):
        ref32 = every.view(np.float16).astype(np.float32)
        for b in range(65536):   # every bit pattern widens exactly
            got = np.float32(lib.h_to_f32(b))
            assert np.isnan(ref32[b]) and np.isnan(got) or got.view(np.uint32) == ref32[b].view(np.uint32), b
        rng = np.random.default_rng(0)
        f32 = np.concatenate([rng.integers(0, 1 << 32, 60000, dtype=np.uint64).astype(np.uint32).view(np.float32),
                              (ref32[rng.integers(0, 65536, 60000)].astype(np.float64) * (1 + rng.uniform(-1e-3, 1e-3, 60000))).astype(np.float32),
                              np.float32([65504, 65519.99, 65520, 65520.01, 6e-8, 5.96e-8, 2.98e-8, 2.9802322e-8, 2.981e-8, 1e-10, 0, -0.0])])
        want = f32.astype(np.float16).view(np.uint16)
        for v, w in zip(f32, want):
            if not np.isnan(v): assert lib.h_from_f32(float(v)) == w, float(v)
        f64 = np.concatenate([rng.uniform(-70000, 70000, 40000), rng.uniform(-1e-4, 1e-4, 20000), rng.uniform(-1e-7, 1e-7, 20000),
                              ref32[rng.integers(0, 65536, 40000)].astype(np.float64) * (1 + rng.uniform(-1e-9, 1e-9, 40000))])
        f64 = f64[np.isfinite(f64)]
        want = f64.astype(np.float16).view(np.uint16)
        for v, w in zip(f64, want): assert lib.h_from_f64(float(v)) == w, float(v)
        a = rng.integers(0, 65536, 40000).astype(np.uint16); b = rng.integers(0, 65536, 40000).astype(np.uint16)
        fa, fb = a.view(np.float16), b.view(np.float16)
        for x, y, m, d in zip(a, b, (fa.astype(np.float32) * fb.astype(np.float32)).astype(np.float16).view(np.uint16),
                              (fa.astype(np.float32) / fb.astype(np.float32)).astype(np.float16).view(np.uint16)):
            fm, fd = np.uint16(m).view(np.float16), np.uint16(d).view(np.float16)
            if not np.isnan(fm): assert lib.h_mul(int(x), int(y)) == m
            if not np.isnan(fd): assert lib.h_div(int(x), int(y)) == d


def test_host_mode_detection_agrees_with_the_oracle_bids(tmp_path):
    """FloatDetect<L>::mult_bid / quant_bid of pco_auto_host.inc on f16, f32 and f64 samples against the oracle's FM<L>::compute_bid /
    quant_compute_bid (two independent restatements of mode/float_mult.rs:62-374 and mode/float_quant.rs:73-149)."""
    lib = _build(tmp_path, "autocmp", f'''
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#include <cmath>
#include <random>
#include "{CSRC}/pco_half.h"
#include "{CSRC}/pco_auto_host.inc"
#include "{ROOT}/oracle/pco_oracle.hpp"
#include "{ROOT}/oracle/pco_oracle_encode.hpp"
using namespace pcogfx::autodetect;
template <class L, class OF> static int run(int n_iter, int* n_found) {{
  typedef typename FloatTraits<L>::F F;
  std::mt19937_64 rng(7);
  int mism = 0; *n_found = 0;
  for (int it = 0; it < n_iter; it++) {{
    const size_t n = 10 + rng() % 3000; const int kind = it % 8;
    const double basef = kind == 0 ? 0.1 : kind == 1 ? 0.01 : kind == 2 ? 3.0 : kind == 3 ? 0.125 : kind == 4 ? (double)(1 + rng() % 50) / 7.0 : kind == 5 ? 0.05 : kind == 6 ? 1e-3 : 0.3;
    const int maxk = 2 + (int)(rng() % (kind == 7 ? 60 : 3000));
    std::vector<F> s; std::vector<OF> so;
    for (size_t i = 0; i < n; i++) {{
      F x = (F)((double)(1 + rng() % maxk) * basef);
      if (rng() % 50 == 0) {{ L b = (L)rng(); std::memcpy(&x, &b, sizeof(L)); }}
      if (kind == 5 && rng() % 3 == 0) {{ L b; std::memcpy(&b, &x, sizeof(L)); b &= ~(L)0x1f; std::memcpy(&x, &b, sizeof(L)); }}
      if (!is_normal<L>(x)) continue;
      const F a = fabsx<L>(x); if (!(a <= max_for_sampling<L>())) continue;
      s.push_back(a); OF o; std::memcpy(&o, &a, sizeof(L)); so.push_back(o);
    }}
    if (s.size() < 10) continue;
    MultConfig<L> c{{}}; double v = 0; const bool ok = FloatDetect<L>::mult_bid(s, c, v);
    pco_oracle::FloatMultConfig<L> oc{{}}; double ov = 0; const bool ook = pco_oracle::FM<L>::compute_bid(so, oc, ov);
    uint32_t k = 0; double qv = 0, oqv = 0; const bool q = FloatDetect<L>::quant_bid(s, k, qv); pco_oracle::Bitlen okk = 0; const bool oq = pco_oracle::FM<L>::quant_compute_bid(so, okk, oqv);
    *n_found += ok;
    if (ok != ook || (ok && (std::memcmp(&c.base, &oc.base, sizeof(L)) || std::memcmp(&c.inv_base, &oc.inv_base, sizeof(L)) || v != ov)) || q != oq || (q && (k != okk || qv != oqv))) mism++;
  }}
  return mism;
}}
extern "C" int cmp16(int n, int* f) {{ return run<uint16_t, pco_oracle::Half>(n, f); }}
extern "C" int cmp32(int n, int* f) {{ return run<uint32_t, float>(n, f); }}
extern "C" int cmp64(int n, int* f) {{ return run<uint64_t, double>(n, f); }}
''')
    for fn, min_found in ((lib.cmp16, 1), (lib.cmp32, 50), (lib.cmp64, 50)):
        found = C.c_int(0)
        assert fn(1200, C.byref(found)) == 0
        assert found.value >= min_found, found.value   # the generator really produces float-mult bids


def test_traffic_tables_are_only_quoted_for_the_build_and_the_kernels_they_were_measured_on(tmp_path, monkeypatch):
    """bench.load_traffic (the `roofline.traffic` of the bench line): a committed PMC table is used only if its csrc_sha16 is the running
    sources' and it lists every kernel that ran; scripts/make_traffic_json.py maps rocprofv3's (truncated) kernel names to the launch labels."""
    root = os.path.join(HERE, "..")
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "scripts"))
    import importlib
    import json
    bench = importlib.import_module("bench")
    prof = tmp_path / "profiles"; prof.mkdir()
    sha = bench.csrc_sha16()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "csrc_sha16", lambda: sha)
    table = {"chunks": 100, "workload": "c2", "csrc_sha16": sha, "kernels": {"enc_pack_kernel": {"fetch_bytes_per_launch": 1000, "write_bytes_per_launch": 500},
                                                                               "dec_walk_kernel<u64>": {"fetch_bytes_per_launch": 300, "write_bytes_per_launch": 0}}}
    (prof / "r09_traffic_c2.json").write_text(json.dumps(table))
    tab, src = bench.load_traffic("c2", 200, ["enc_pack_kernel", "dec_walk_kernel<u64>"])
    assert src == "r09_traffic_c2.json" and tab["enc_pack_kernel"] == 3000 and tab["dec_walk_kernel<u64>"] == 600   # scaled by the chunk count
    tab, src = bench.load_traffic("c2", 200, ["enc_pack_kernel", "enc_walkd_kernel"])
    assert tab is None and "enc_walkd_kernel" in src
    table["csrc_sha16"] = "0" * 16
    (prof / "r09_traffic_c2.json").write_text(json.dumps(table))
    tab, src = bench.load_traffic("c2", 200, ["enc_pack_kernel"])
    assert tab is None and "other kernel sources" in src
    tab, src = bench.load_traffic("c9", 200, ["enc_pack_kernel"])
    assert tab is None
    # the label mapping (names as pmc_summary.py prints them: cut at 34 characters)
    src_text = open(os.path.join(root, "scripts", "make_traffic_json.py")).read()
    ns = {"re": __import__("re")}
    exec(src_text[src_text.index("TY = {"):src_text.index("def parse(")], ns)
    label = ns["label"]
    assert label("dec_walk_kernel<unsigned long, 8u,") == "dec_walk_kernel<u64>" and label("dec_walk_kernel<unsigned int, 4u, ") == "dec_walk4_kernel<u32>"
    assert label("dec_expand_kernel<unsigned long, f") == "dec_expand_kernel<u64>" and label("dec_expand_kernel<unsigned long, t") == "dec_expand_lb_kernel<u64>"
    assert label("dec_walk_trail_kernel<unsigned lon") == "dec_walk_kernel<u64>" and label("dec_trail_kernel<unsigned long>") == "dec_trail_kernel<u64>" and label("dec_trail_kernel<unsigned int>") == "dec_trail_kernel<u32>" and label("dec_trail_kernel<unsigned long, tr") == "dec_trail2_kernel<u64>" and label("dec_trail_kernel<unsigned long, fa") == "dec_trail_kernel<u64>" and label("dec_trail_kernel<unsigned short, t") == "dec_trail2_kernel<u16>"
    assert label("enc_walk_kernel<16u>") == "enc_walk16_kernel" and label("enc_walk_kernel<8u>") == "enc_walk_kernel" and label("enc_walkd_kernel") == "enc_walkd_kernel"
    assert label("enc_lookback_kernel<LbCfg<256u, 25") == "enc_lookback_kernel<small>" and label("enc_lookback_kernel<LbCfg<1024u, 1") == "enc_lookback_kernel"
    assert label("enc_lookback_pipe_kernel<LbPipe<fa") == "enc_lookback_pipe_kernel" and label("enc_hist_select_kernel<unsigned in") == "enc_hist_select_kernel"
    assert label("enc_split_kernel<true, false>") == "enc_split_kernel<c16>" and label("pco_decode_kernel<unsigned long>") == "pco_decode_kernel<u64>"


def test_chunk_meta_accessor_reads_what_the_oracle_wrote():
    """include/pco_gfx.h PcoGfxChunkMetaInfo (ChunkCompressor::meta / ChunkDecompressor::meta): a host-side parse of the ChunkMeta bytes --
    no device involved -- compared with the oracle's own reading of the same chunks; cut metadata is InsufficientData."""
    import ctypes as C
    import oracle_lib as O
    from pcodec_amd import _lib as G

    class Info(C.Structure):
        _fields_ = [("mode_kind", C.c_uint32), ("mode_k", C.c_uint32), ("mode_base_latent", C.c_uint64), ("delta_kind", C.c_uint32), ("delta_order", C.c_uint32),
                    ("window_n_log", C.c_uint32), ("state_n_log", C.c_uint32), ("secondary_uses_delta", C.c_uint32), ("n_vars_parsed", C.c_uint32),
                    ("present", C.c_uint32 * 3), ("ans_size_log", C.c_uint32 * 3), ("n_bins", C.c_uint32 * 3), ("meta_bytes", C.c_uint64)]
    L = G.lib()
    rng = np.random.default_rng(8)
    ramp = (np.uint64(1 << 40) + np.uint64(1000) * np.arange(9000, dtype=np.uint64) + rng.integers(0, 512, 9000).astype(np.uint64))
    season = (rng.integers(-(1 << 40), 1 << 40, 365)[np.arange(9000) % 365] + rng.integers(-3, 4, 9000)).astype(np.int64)
    cases = [(ramp, dict(mode=1, delta=2, delta_order=1)), (ramp, dict(mode=1, delta=2, delta_order=3)), (rng.integers(1000, 10000, 9000) / 100.0, dict(mode=2, mode_f64=0.01, delta=1)),
             (season, dict(mode=1, delta=3)), ((rng.integers(0, 5000, 9000) * 77).astype(np.uint32), dict(mode=4, mode_u64=77, delta=1)),
             (rng.standard_normal(9000).astype(np.float32), dict(mode=3, mode_u64=8, delta=1)), (ramp, dict())]
    for nums, kw in cases:
        meta, pages, _ = O.wrapped_compress(nums, O.make_config(**kw))
        whole = O.simple_compress(nums, O.make_config(**kw))
        want, _ = O.inspect_first_chunk(whole)
        got = Info()
        mb = np.frombuffer(meta, np.uint8)
        assert L.pco_gfx_chunk_meta_info(mb.ctypes.data_as(C.c_void_p), C.c_size_t(len(meta)), C.c_ubyte(G.DTYPE_BYTE[nums.dtype.name]), C.c_uint8(4), C.byref(got)) == 0, kw
        assert (got.mode_kind, got.mode_k, got.mode_base_latent, got.delta_kind, got.delta_order) == (want.mode_kind, want.mode_k, want.mode_base_latent, want.delta_kind, want.delta_order), kw
        if got.delta_kind == 2:
            assert (got.window_n_log, got.state_n_log) == (want.window_n_log, want.state_n_log)
        assert got.n_vars_parsed == 3 and got.meta_bytes == len(meta)
        assert list(got.present) == [int(bool(x)) for x in want.var_present] and list(got.n_bins) == list(want.n_bins) and list(got.ans_size_log) == list(want.ans_size_log), kw
        for cut in (0, 1, len(meta) // 2, len(meta) - 1):
            assert L.pco_gfx_chunk_meta_info(mb.ctypes.data_as(C.c_void_p), C.c_size_t(cut), C.c_ubyte(G.DTYPE_BYTE[nums.dtype.name]), C.c_uint8(4), C.byref(got)) != 0
            assert L.pco_gfx_last_status() == G.ST_INSUFFICIENT_DATA, (kw, cut)


def test_chunk_meta_accessor_rejects_what_the_reference_rejects():
    """metadata/delta_encoding.rs:143-176: a Consecutive order of 0, a lookback window log beyond MAX_DELTA_LOOKBACK_WINDOW_N_LOG (24) and a
    state log beyond the window log are Corruption in ChunkMeta::read_from -- and here (round 4 returned PcoSuccess for all three)."""
    import ctypes as C
    from pcodec_amd import _lib as G
    L = G.lib()
    out = (C.c_uint8 * 256)()

    def bits(fields):   # little-endian bit stream
        v = 0; n = 0
        for val, w in fields:
            v |= (val & ((1 << w) - 1)) << n; n += w
        return np.frombuffer(int(v).to_bytes((n + 7) // 8 + 16, "little"), np.uint8).copy()
    cases = {"order 0": [(0, 4), (1, 4), (0, 3), (0, 1)],
             "window 25": [(0, 4), (2, 4), (24, 5), (0, 4), (0, 1)],
             "state beyond window": [(0, 4), (2, 4), (3, 5), (9, 4), (0, 1)]}
    for name, f in cases.items():
        b = bits(f)
        assert L.pco_gfx_chunk_meta_info(b.ctypes.data_as(C.c_void_p), C.c_size_t(len(b)), C.c_ubyte(1), C.c_uint8(4), C.byref(out)) != 0, name
        assert L.pco_gfx_last_status() == G.ST_CORRUPTION, name
    ok = bits([(0, 4), (2, 4), (23, 5), (4, 4), (0, 1), (7, 4), (0, 15), (8, 4), (0, 15)])   # window 24, state 4: the largest the reference accepts
    assert L.pco_gfx_chunk_meta_info(ok.ctypes.data_as(C.c_void_p), C.c_size_t(len(ok)), C.c_ubyte(1), C.c_uint8(4), C.byref(out)) == 0
