"""One encode call over k copies of the same chunk: every output must be the same bytes (and the oracle's).  Hunts timing-dependent faults.
usage: [PCO_GFX_LIB=...] enc_consistency.py <workload kind> <chunks> [reps]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import gpu_util as U, oracle_lib as O
from pcodec_amd import _lib as G
kind, k = sys.argv[1], int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
L = G.lib()
nums = U.synth(kind); gcfg, ocfg = U.cfg_pair(kind)
want = np.frombuffer(O.simple_compress(nums, ocfg), np.uint8)
src = torch.from_numpy(nums.view(np.uint8).reshape(-1).copy()).cuda().repeat(k).contiguous()
cap = (L.pco_gfx_guarantee_chunk_size(nums.size, G.DTYPE_BYTE[nums.dtype.name]) + 64 + 15) // 16 * 16
dst = torch.zeros(cap * k, dtype=torch.uint8, device="cuda")
tasks = (G.EncodeTask * k)(*[G.EncodeTask(src.data_ptr() + i * nums.nbytes, nums.size, dst.data_ptr() + i * cap, cap, G.DTYPE_BYTE[nums.dtype.name], 0) for i in range(k)])
res = (G.TaskResult * k)()
body = torch.from_numpy(want[9:len(want) - 1].copy()).cuda()   # the chunk without the 9-byte... compared loosely below
for rep in range(reps):
    dst.zero_()
    code = L.pco_gfx_compress_chunks(k, tasks, C.byref(gcfg), res, None, None); torch.cuda.synchronize()
    sizes = np.array([res[i].n_out for i in range(k)]); st = np.array([res[i].status for i in range(k)])
    n0 = int(sizes[0])
    view = dst.view(k, cap)[:, :n0]
    bad = (view != view[0:1]).any(dim=1).nonzero().flatten().cpu().numpy()
    ref_ok = bool((view[0].cpu().numpy() == want[len(want) - 1 - n0:len(want) - 1]).all()) if n0 <= len(want) else False
    print(kind, k, "rep", rep, "code", code, "sizes equal", bool((sizes == n0).all()), "status ok", bool((st == 0).all()), "chunks differing from chunk 0:", len(bad), bad[:8], "chunk 0 == oracle chunk:", ref_ok)
