"""Where and when the blocks of enc_walkseg_kernel ran (needs a -DPCO_WS_TRACE build: PCO_GFX_LIB=ab/libpco_gfx_wstrace.so).  usage: ws_trace.py <chunks>"""
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
for rep in range(2): L.pco_gfx_compress_chunks(k, tasks, C.byref(gcfg), res, None, None); torch.cuda.synchronize()
out = np.zeros(3 * k, np.uint64); L.pco_gfx_debug_ws_trace(out.ctypes.data_as(C.c_void_p), k)
t = out.reshape(-1, 3); hw = t[:, 0].astype(np.int64); st = t[:, 1].astype(np.int64); en = t[:, 2].astype(np.int64)
t0 = st.min(); dur = (en - st) / 100.0   # us
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; xcc = np.arange(k) % 8   # (HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]; blocks go round the XCDs)
key = xcc * 10000 + se * 100 + sh * 16 + cu
print("blocks", k, "kernel span us", (en.max() - t0) / 100.0, "block duration us: median %.1f min %.1f max %.1f" % (np.median(dur), dur.min(), dur.max()), "distinct CUs", len(set(key.tolist())))
# concurrency on the busiest CU: blocks resident at the median time
mid = (t0 + en.max()) // 2
res_mid = {}
for kk, a, b in zip(key, st, en):
    if a <= mid < b: res_mid[kk] = res_mid.get(kk, 0) + 1
v = np.array(list(res_mid.values()) or [0]); print("resident blocks per CU at mid-kernel: mean %.2f max %d over %d CUs" % (v.mean(), v.max(), len(v)))
