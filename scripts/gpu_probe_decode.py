"""First-contact probe: decode oracle-compressed chunks on the GPU, check parity, time a many-chunk batch."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle_lib as O
import gpu_util as U
from pcodec_amd import _lib as G
import torch

L = G.lib()
print("devices", L.pco_gfx_device_count(), flush=True)
ok_all = True
# 1. golden assets
from test_oracle_golden import EXPECTED, asset
for name, exp in sorted(EXPECTED.items()):
    if exp.dtype.itemsize == 1:
        continue
    try:
        got = U.gpu_simple_decompress(asset(name), exp.dtype, max(exp.size, 1))
        ok = U.bits_equal(got, exp)
    except Exception as e:  # noqa
        ok = False; print("  ", e)
    print("asset", name, ok, flush=True); ok_all &= ok
# 2. configs
for kind in ["c1", "c2", "c3", "c3d", "c4"]:
    nums = U.synth(kind)
    _, ocfg = U.cfg_pair(kind)
    t0 = time.time(); enc = O.simple_compress(nums, ocfg); t1 = time.time()
    try:
        got = U.gpu_simple_decompress(enc, nums.dtype, nums.size)
        ok = U.bits_equal(got, nums)
    except Exception as e:  # noqa
        ok = False; print("  ", e)
    print(f"config {kind}: parity {ok}  compressed {len(enc)} B ({8*len(enc)/nums.size:.2f} bits/elem) oracle enc {t1-t0:.3f}s", flush=True)
    ok_all &= ok
# 3. small / ragged
rng = np.random.default_rng(5)
for dt in [np.uint32, np.int32, np.uint64, np.int64, np.float32, np.float64, np.uint16, np.int16]:
    for n in [1, 2, 3, 255, 256, 257, 1000, 4099]:
        for (dk, do) in [(1, 0), (2, 1), (2, 2), (2, 7), (3, 0), (0, 0)]:
            if np.dtype(dt).kind == "f": nums = (rng.standard_normal(n) * 100).astype(dt)
            else:
                ii = np.iinfo(dt); nums = rng.integers(max(ii.min, -(1 << 40)), min(ii.max, 1 << 40), n).astype(dt)
            enc = O.simple_compress(nums, O.make_config(delta=dk, delta_order=do))
            try:
                got = U.gpu_simple_decompress(enc, nums.dtype, nums.size)
                ok = U.bits_equal(got, nums)
            except Exception as e:  # noqa
                ok = False; print("  ", e)
            if not ok: print("FAIL", dt, n, dk, do, flush=True)
            ok_all &= ok
print("small cases done", ok_all, flush=True)
# 4. batched decode timing (c2)
for kind, nchunks in [("c2", 1024), ("c2", 4096), ("c3", 2048)]:
    nums = U.synth(kind); _, ocfg = U.cfg_pair(kind)
    enc = O.simple_compress(nums, ocfg)
    chunk = enc[10:-1]  # strip file header (10 B) and footer
    stride = (len(chunk) + 64 + 15) // 16 * 16
    src = torch.zeros(nchunks * stride, dtype=torch.uint8, device="cuda")
    one = torch.frombuffer(bytearray(chunk), dtype=torch.uint8).cuda()
    src.view(nchunks, stride)[:, : len(chunk)] = one
    dst = torch.empty(nchunks * nums.size, dtype=torch.int64 if nums.dtype.itemsize == 8 else torch.int32, device="cuda")
    tasks = (G.DecodeTask * nchunks)()
    dtb = G.DTYPE_BYTE[nums.dtype.name]
    for i in range(nchunks):
        tasks[i] = G.DecodeTask(src.data_ptr() + i * stride, len(chunk), dst.data_ptr() + i * nums.nbytes, nums.size, dtb, 0)
    res = (G.TaskResult * nchunks)()
    torch.cuda.synchronize()
    for it in range(3):
        t0 = time.time()
        code = L.pco_gfx_decompress_chunks(nchunks, tasks, res, None, None)
        torch.cuda.synchronize(); t1 = time.time()
        G.check(code)
        print(f"batched decode {kind} x{nchunks}: {t1-t0:.4f}s  {nchunks*nums.nbytes/(t1-t0)/1e9:.1f} GB/s", flush=True)
    out = dst.view(nchunks, -1)[nchunks - 1].cpu().numpy().view(nums.dtype)
    ok = U.bits_equal(out, nums); ok_all &= ok
    print("  last chunk parity", ok, "status", res[nchunks - 1].status, res[0].n_out, flush=True)
print("ALL OK" if ok_all else "SOME FAILED")
