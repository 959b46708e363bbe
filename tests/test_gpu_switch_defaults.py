"""Every production switch of libpco_gfx.so has the default its documentation claims when the environment is clean.

Why this exists: the switches are read from the environment once, in initialisers; in round 5 one such initialiser (a lambda) was compiled
with another lambda's body and every call silently ran strict histograms.  The initialisers are named functions now (pco_host.h); this test
pins what they must evaluate to by looking at which kernels a call launches (pco_gfx_profile_begin / _end) in a child process whose
environment holds no PCO_GFX_* variable at all."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
pytestmark = pytest.mark.gpu

CHILD = r'''
import ctypes as C, json, sys
import numpy as np
sys.path.insert(0, '@ROOT@'); sys.path.insert(0, '@HERE@')
import torch
from pcodec_amd import _lib as G
import gpu_util as U
L = G.lib()

def kernels_of(fn):
    L.pco_gfx_profile_begin()
    fn()
    names = C.create_string_buffer(1 << 16); ms = (C.c_float * 4096)()
    nk = L.pco_gfx_profile_end(names, len(names), ms, 4096)
    raw = names.raw; out = []; pos = 0
    for _ in range(nk):
        e = raw.index(b"\0", pos); out.append(raw[pos:e].decode()); pos = e + 1
    return sorted(set(out))

def batch(arrays, cfg):
    return lambda: U.gpu_batched(arrays, cfg)

rep = {}
ramp = lambda n, s: (np.uint64(1 << 40) + np.uint64(1000) * np.arange(n, dtype=np.uint64) + np.random.default_rng(s).integers(0, 512, n).astype(np.uint64))
c2 = G.make_config(mode=1, delta=2, delta_order=1)
# a handful of long chunks: the segmented encode walk (<= 4096 items), no expanders under the decode walk (< 1024 chunks)
rep["few_long"] = kernels_of(batch([ramp(1 << 15, s) for s in range(8)], c2))
# 1100 short chunks: the expanders under the walk (>= 1024 chunks of one width); pages below 64 batches never take the segmented walk
rep["many_short"] = kernels_of(batch([ramp(2048, s) for s in range(1100)], c2))
# 4200 chunks of 64 batches: more than 4096 items -> the unsegmented encode walkers
rep["over_4096_items"] = kernels_of(batch([ramp(1 << 14, s % 7) for s in range(4200)], c2))
# lookback on full pages: hash pre-pass + pipeline
rep["lookback"] = kernels_of(batch([U.synth("c4", 1 << 16, seed=s) for s in range(4)], G.make_config(mode=1, delta=3)))
rep["strict_fallbacks"] = int(L.pco_gfx_strict_histogram_fallbacks())
rep["givebacks"] = int(L.pco_gfx_trail_givebacks()); rep["marked"] = int(L.pco_gfx_trail_marked())
# unknown flag bits are refused
bad = G.make_config(mode=1, delta=1); bad.flags = 0x80
x = ramp(1000, 1); dst = np.zeros(1 << 16, np.uint8); n = C.c_size_t(0)
code = L.pco_gfx_simple_compress_into_ex(x.ctypes.data_as(C.c_void_p), x.size, 2, C.byref(bad), 0, dst.ctypes.data_as(C.c_void_p), dst.size, C.byref(n))
rep["unknown_flag"] = [int(code), int(L.pco_gfx_last_status())]
print("REPORT " + json.dumps(rep))
'''


def test_switch_defaults_in_a_clean_environment():
    env = {k: v for k, v in os.environ.items() if not k.startswith("PCO_GFX_")}
    r = subprocess.run([sys.executable, "-c", CHILD.replace("@ROOT@", ROOT).replace("@HERE@", HERE)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith("REPORT ")][-1][7:])
    every = set().union(*[set(rep[k]) for k in ("few_long", "many_short", "over_4096_items", "lookback")])
    # strict histograms are opt-in (PCO_GFX_STRICT_HISTOGRAM / PCO_GFX_CFG_STRICT_HISTOGRAM)
    assert not any("enc_hist_literal" in k for k in every) and rep["strict_fallbacks"] == 0
    # the fast paths are on (PCO_GFX_NO_FAST_DECODE / _NO_FAST_ENCODE / _NO_WALKD / _NO_C16 unset)
    assert any(k.startswith("dec_walk") for k in rep["few_long"]) and "enc_pack_kernel" in rep["few_long"] and "enc_split_kernel<c16>" in rep["few_long"]
    # PCO_GFX_WALK_SEG default 1: segmented encode walk for calls of up to 4096 items with pages of >= 64 batches, and only there
    assert "enc_walkseg_kernel" in rep["few_long"]
    assert "enc_walkseg_kernel" not in rep["many_short"] and "enc_walkseg_kernel" not in rep["over_4096_items"]
    assert "enc_walkd_kernel" in rep["over_4096_items"]
    # PCO_GFX_WALKP / PCO_GFX_PLACE_FORK default on: walk + pack in one block where the walk is not segmented, its bodies moved into place on the second stream
    assert "enc_walkp_kernel" in rep["over_4096_items"] and "~enc_place_kernel" in rep["over_4096_items"] and "enc_place+pack" in rep["over_4096_items"]
    assert "enc_walkp_kernel" not in rep["few_long"]
    # PCO_GFX_DEC_TRAIL default 1 (not 2): the expanders under the walk from 1024 chunks of one width on, not below
    assert any(k.startswith("dec_walk+trail") for k in rep["many_short"]) and any(k.startswith("~dec_trail_kernel") for k in rep["many_short"])
    assert not any("trail" in k for k in rep["few_long"])
    assert rep["marked"] >= 1100 and rep["givebacks"] == 0      # PCO_GFX_TRAIL_DEBUG unset: nobody times out, nothing is given back
    # PCO_GFX_LB_PIPE / _LB_PROPS / _LB_FASTD default on: hash pre-pass + the pipeline's fast stage D
    assert "enc_lookback_hash_kernel" in rep["lookback"] and "enc_lookback_pipe_kernel<props,fastd>" in rep["lookback"]
    # unknown PcoChunkConfigEx::flags bits: PcoCompressionError / InvalidArgument
    assert rep["unknown_flag"] == [2, 3]
