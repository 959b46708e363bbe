"""Phase times inside enc_walkp_kernel (a -DPCO_WP_TIMING build): one packing wave and the walker of block 7, cycles per batch step.
usage: PCO_GFX_LIB=ab/libpco_gfx_wpt.so python scripts/wp_timing.py <chunks>"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import gpu_util as U
from pcodec_amd import _lib as G
k = int(sys.argv[1]); L = G.lib()
nums = U.synth("c2"); gcfg, _ = U.cfg_pair("c2")
src = torch.from_numpy(nums.view(np.uint8).reshape(-1).copy()).cuda().repeat(k).contiguous()
cap = (L.pco_gfx_guarantee_chunk_size(nums.size, G.DTYPE_BYTE[nums.dtype.name]) + 64 + 15) // 16 * 16
dst = torch.zeros(cap * k, dtype=torch.uint8, device="cuda")
tasks = (G.EncodeTask * k)(*[G.EncodeTask(src.data_ptr() + i * nums.nbytes, nums.size, dst.data_ptr() + i * cap, cap, G.DTYPE_BYTE[nums.dtype.name], 0) for i in range(k)])
res = (G.TaskResult * k)()
for rep in range(2):
    L.pco_gfx_compress_chunks(k, tasks, C.byref(gcfg), res, None, None); torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)(); L.pco_gfx_debug_wp_timing(buf)
n = max(buf[5], 1)
print("per step (cycles of s_memtime, 100 MHz x?):", "find %.0f gather+load %.0f puts %.0f flush %.0f barrier %.0f | walker: walk %.0f barrier %.0f | steps %d" % (buf[0] / n, buf[1] / n, buf[2] / n, buf[3] / n, buf[4] / n, buf[8] / n, buf[9] / n, n))
