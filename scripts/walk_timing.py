"""Cycle breakdown of dec_walk_kernel's rounds (block 0) for one or more A/B builds of the library.

Each build must be compiled with -DPCO_WALK_TIMING (scripts/build_variant.sh <name> -DPCO_WALK_TIMING [...]).  One process
per build (a process can load one copy of the library); timing-only variants may decode garbage, so the decode status is
not checked.  usage: walk_timing.py <chunks> <workload> <lib> [<lib> ...]   (or a single run with PCO_GFX_LIB set)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
wl = sys.argv[2] if len(sys.argv) > 2 else "c2"
libs = sys.argv[3:]
if libs:
    for lib in libs:
        env = dict(os.environ, PCO_GFX_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, __file__, str(chunks), wl], env=env, capture_output=True, text=True)
        print(os.path.basename(lib).ljust(34), (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], flush=True)
    sys.exit(0)

import numpy as np
import torch
import bench as B
from pcodec_amd import _lib as G

L = G.lib()
dev = torch.device("cuda", 0)
tdt, dtb, cfg_kw, desc = B.WORKLOADS[wl]
gcfg = G.make_config(**cfg_kw)
data = B.make_chunks(torch, wl, chunks, 0, dev)
cb = B.N18 * B.ELEM_BYTES[wl]
cap = (L.pco_gfx_guarantee_chunk_size(B.N18, dtb) + 64 + 15) // 16 * 16
comp = torch.zeros(chunks * cap, dtype=torch.uint8, device=dev)
out = torch.empty_like(data)
et = np.zeros(chunks, B.ENC_TASK)
et["src"] = data.data_ptr() + np.arange(chunks, dtype=np.uint64) * cb
et["n"] = B.N18; et["dtype"] = dtb; et["dst_cap"] = cap
et["dst"] = comp.data_ptr() + np.arange(chunks, dtype=np.uint64) * cap
dt = np.zeros(chunks, B.DEC_TASK)
dt["src"] = et["dst"]; dt["dst"] = out.data_ptr() + np.arange(chunks, dtype=np.uint64) * cb
dt["dst_cap"] = B.N18; dt["dtype"] = dtb
er = np.zeros(chunks, B.RESULT); dr = np.zeros(chunks, B.RESULT)
G.check(L.pco_gfx_compress_chunks(chunks, et.ctypes.data, C.byref(gcfg), er.ctypes.data, None, None))
dt["src_len"] = er["n_out"]
for _ in range(2):
    L.pco_gfx_decompress_chunks(chunks, dt.ctypes.data, dr.ctypes.data, None, None)   # status deliberately ignored
torch.cuda.synchronize()
ok = bool(torch.equal(out, data))
buf = (C.c_ulonglong * 8)()
L.pco_gfx_debug_walk_timing(buf)
stage, walk, tail, rounds, t0, t1 = list(buf)[:6]
s1 = buf[6]; s2 = buf[7] & 0xffffffff; s3 = buf[7] >> 32
print(f"rounds {rounds} [pre-load {s1 / max(rounds, 1):.0f} loads {s2 / max(rounds, 1):.0f} lds-write+sync {s3 / max(rounds, 1):.0f}] stage {stage / max(rounds, 1):.0f} walk {walk / max(rounds, 1):.0f} tail {tail / max(rounds, 1):.0f} total {t1 - t0} roundtrip_ok {ok}")
