"""Phase breakdown of dec_trail_kernel (block 0, wave 0) for a build with -DPCO_TRAIL_TIMING: iterations, then s_memtime units per
iteration spent evaluating the progress words, in stage A, issuing the requests, in stage B.  usage: trail_timing.py <chunks> [workload]
(PCO_GFX_LIB selects the build; PCO_GFX_TRAIL_DEBUG=s times the expanders alone, after the walk)."""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench as B
from pcodec_amd import _lib as G

chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
wl = sys.argv[2] if len(sys.argv) > 2 else "c2"
L = G.lib(); dev = torch.device("cuda", 0)
kinds, cfg_kw, _, _ = B.WORKLOADS[wl]
kind = kinds[0]; npdt, dtb = B.KINDS[kind]
g = torch.Generator(device=dev); g.manual_seed(1)
data = B.make_kind(torch, kind, chunks, g, dev).contiguous()
cb = B.N18 * np.dtype(npdt).itemsize
cap = (L.pco_gfx_guarantee_chunk_size(B.N18, dtb) + 64 + 15) // 16 * 16
comp = torch.zeros(chunks * cap, dtype=torch.uint8, device=dev); out = torch.empty_like(data)
et = np.zeros(chunks, B.ENC_TASK)
et["src"] = data.data_ptr() + np.arange(chunks, dtype=np.uint64) * cb; et["n"] = B.N18; et["dtype"] = dtb; et["dst_cap"] = cap
et["dst"] = comp.data_ptr() + np.arange(chunks, dtype=np.uint64) * cap
dt = np.zeros(chunks, B.DEC_TASK)
dt["src"] = et["dst"]; dt["dst"] = out.data_ptr() + np.arange(chunks, dtype=np.uint64) * cb; dt["dst_cap"] = B.N18; dt["dtype"] = dtb
er = np.zeros(chunks, B.RESULT); dr = np.zeros(chunks, B.RESULT)
gcfg = G.make_config(**cfg_kw)
G.check(L.pco_gfx_compress_chunks(chunks, et.ctypes.data, C.byref(gcfg), er.ctypes.data, None, None))
dt["src_len"] = er["n_out"]
for _ in range(2):
    L.pco_gfx_decompress_chunks(chunks, dt.ctypes.data, dr.ctypes.data, None, None)
torch.cuda.synchronize()
ok = bool(torch.equal(out.view(torch.uint8), data.view(torch.uint8)))
buf = (C.c_ulonglong * 8)()
L.pco_gfx_debug_trail_timing(buf)
it = max(int(buf[0]), 1)
print(f"iterations {buf[0]} (idle {buf[5]}): per iteration -- poll {buf[1] / it:.0f} stageA {buf[2] / it:.0f} requests {buf[3] / it:.0f} stageB {buf[4] / it:.0f} | roundtrip_ok {ok}")

# per-block start / end stamps of the two kernels (device-wide 100 MHz clock): how far behind their walkers the expanders end
st = np.zeros((4, 4096), np.uint64)
if L.pco_gfx_debug_trail_stamps(st.ctypes.data_as(C.c_void_p)) == 0:
    nb = (chunks + 7) // 8
    s = st[:, :min(nb, 4096)].astype(np.int64)
    t0 = s[s > 0].min()
    us = lambda a: (a - t0) / 100.0
    q = lambda a: " ".join(f"{np.percentile(a, p):8.0f}" for p in (0, 10, 50, 90, 100))
    print("microseconds from the first stamp; percentiles 0 10 50 90 100 over the blocks")
    for name, row in zip(("walker start", "walker end", "expanders start", "expanders end"), s):
        print(f"  {name:16s} {q(us(row))}")
    print(f"  {'end lag':16s} {q((s[3] - s[1]) / 100.0)}")
    print(f"  {'walker duration':16s} {q((s[1] - s[0]) / 100.0)}")
