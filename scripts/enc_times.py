"""Per-kernel times of one batched ENCODE call (no decode, no verification: for ablation builds that write wrong bytes on purpose).
usage: [PCO_GFX_LIB=...] enc_times.py <workload kind of tests/gpu_util.synth> <chunks> [filter]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import gpu_util as U
from pcodec_amd import _lib as G

kind, k = sys.argv[1], int(sys.argv[2]); flt = sys.argv[3] if len(sys.argv) > 3 else ""
L = G.lib()
data_kind, cfg_kind = (kind.split("+") + [kind])[:2] if "+" in kind else (kind, kind)   # e.g. c2+auto: the c2 data under the default ChunkConfig
nums = U.synth(data_kind); gcfg, _ = U.cfg_pair(cfg_kind)
src = torch.from_numpy(nums.view(np.uint8).reshape(-1).copy()).cuda()
srcs = src.repeat(k).contiguous()
cap = (L.pco_gfx_guarantee_chunk_size(nums.size, G.DTYPE_BYTE[nums.dtype.name]) + 64 + 15) // 16 * 16
dst = torch.zeros(cap * k, dtype=torch.uint8, device="cuda")
tasks = (G.EncodeTask * k)(*[G.EncodeTask(srcs.data_ptr() + i * nums.nbytes, nums.size, dst.data_ptr() + i * cap, cap, G.DTYPE_BYTE[nums.dtype.name], 0) for i in range(k)])
res = (G.TaskResult * k)()
for rep in range(3):
    L.pco_gfx_profile_begin()
    code = L.pco_gfx_compress_chunks(k, tasks, C.byref(gcfg), res, None, None)
    torch.cuda.synchronize()
    names = C.create_string_buffer(1 << 16); ms = (C.c_float * 4096)()
    nk = L.pco_gfx_profile_end(names, len(names), ms, 4096)
raw = names.raw; nm = []; pos = 0
for _ in range(nk):
    e = raw.index(b"\0", pos); nm.append(raw[pos:e].decode()); pos = e + 1
tot = {}
for a, b in zip(nm, ms[:nk]): tot[a] = tot.get(a, 0.0) + b
print(kind, k, "code", code, " ".join(f"{a}={b:.3f}" for a, b in sorted(tot.items()) if flt in a), "sum=%.3f" % sum(tot.values()))
