"""Probe: GPU encode vs oracle bytes, then batched encode/decode timing."""
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
ok_all = True


def compare(name, nums, gcfg, ocfg):
    global ok_all
    want = O.simple_compress(nums, ocfg)
    try:
        got = U.gpu_simple_compress(nums, gcfg)
    except Exception as e:  # noqa
        print(f"{name}: GPU ERROR {e}", flush=True); ok_all = False; return
    if got == want:
        print(f"{name}: MATCH ({len(got)} B)", flush=True); return
    ok_all = False
    k = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), min(len(got), len(want)))
    print(f"{name}: DIFF len got {len(got)} want {len(want)} first diff at byte {k}", flush=True)
    try:
        gi, gb = O.inspect_first_chunk(got); wi, wb = O.inspect_first_chunk(want)
        print("   got : mode", gi.mode_kind, "delta", gi.delta_kind, gi.delta_order, "asl", list(gi.ans_size_log), "nbins", list(gi.n_bins), "meta_end", gi.meta_end_byte)
        print("   want: mode", wi.mode_kind, "delta", wi.delta_kind, wi.delta_order, "asl", list(wi.ans_size_log), "nbins", list(wi.n_bins), "meta_end", wi.meta_end_byte)
        for v in range(3):
            if len(gb[v]) != len(wb[v]) or not np.array_equal(gb[v], wb[v]):
                m = min(len(gb[v]), len(wb[v]))
                d = next((i for i in range(m) if not np.array_equal(gb[v][i], wb[v][i])), m)
                print(f"   var {v} bins differ first at {d}: got {gb[v][max(0,d-1):d+2].tolist()} want {wb[v][max(0,d-1):d+2].tolist()}")
        dec = O.simple_decompress(got, nums.dtype, cap=nums.size + 8)
        print("   GPU output decodes losslessly:", U.bits_equal(dec, nums))
    except Exception as e:  # noqa
        print("   inspect/decode failed:", e)


for kind in ["c2", "c1", "c3", "c3d"]:
    g, o = U.cfg_pair(kind)
    compare(f"config {kind}", U.synth(kind), g, o)

rng = np.random.default_rng(11)
cases = []
for dt in [np.uint32, np.int32, np.uint64, np.int64, np.float32, np.float64]:
    for n in [1, 2, 3, 7, 255, 256, 257, 1000, 4099, 70000]:
        for (dk, do) in [(1, 0), (2, 1), (2, 2), (2, 7)]:
            for dist in range(3):
                if np.dtype(dt).kind == "f":
                    nums = [(rng.standard_normal(n) * 100), rng.integers(0, 50, n) * 0.5, np.cumsum(rng.standard_normal(n))][dist].astype(dt)
                else:
                    ii = np.iinfo(dt)
                    nums = [rng.integers(max(ii.min, -(1 << 40)), min(ii.max, 1 << 40), n), rng.integers(0, 20, n),
                            np.cumsum(rng.integers(-5, 100, n))][dist].astype(dt)
                cases.append((f"{np.dtype(dt).name} n={n} delta=({dk},{do}) dist={dist}", nums, dict(mode=1, delta=dk, delta_order=do)))
nbad = 0
for name, nums, kw in cases:
    want = O.simple_compress(nums, O.make_config(**kw))
    try:
        got = U.gpu_simple_compress(nums, G.make_config(enable_8_bit=True, **kw))
    except Exception as e:  # noqa
        got = b""; print(name, "GPU ERROR", e, flush=True)
    if got != want:
        nbad += 1
        if nbad <= 8:
            compare(name, nums, G.make_config(enable_8_bit=True, **kw), O.make_config(**kw))
print(f"classic matrix: {len(cases) - nbad}/{len(cases)} byte-identical", flush=True)
ok_all &= nbad == 0
# float modes
for dt in (np.float32, np.float64):
    x = (rng.integers(1000, 10000, 5000) / 100.0).astype(dt)
    compare(f"{np.dtype(dt).name} float_mult 0.01", x, G.make_config(mode=2, mode_f64=0.01, delta=1), O.make_config(mode=2, mode_f64=0.01, delta=1))
    compare(f"{np.dtype(dt).name} float_quant", x.astype(np.float32).astype(dt) if dt == np.float64 else x, G.make_config(mode=3, mode_u64=12, delta=1), O.make_config(mode=3, mode_u64=12, delta=1))
for dt in (np.uint32, np.int64):
    x = (rng.integers(0, 5000, 5000) * 8 + rng.integers(0, 2, 5000)).astype(dt)
    compare(f"{np.dtype(dt).name} int_mult 8", x, G.make_config(mode=4, mode_u64=8, delta=1), O.make_config(mode=4, mode_u64=8, delta=1))
# fallback: incompressible data with delta
x = rng.integers(0, 1 << 63, 3000, dtype=np.uint64)
compare("fallback u64 delta1", x, G.make_config(mode=1, delta=2, delta_order=1), O.make_config(mode=1, delta=2, delta_order=1))

# ---- batched timing ----
for kind, nchunks in [("c2", 2048), ("c3", 1024), ("c1", 1024)]:
    nums = U.synth(kind); gcfg, ocfg = U.cfg_pair(kind)
    dtb = G.DTYPE_BYTE[nums.dtype.name]
    src = torch.from_numpy(nums.view(np.int64 if nums.itemsize == 8 else np.int32)).cuda().repeat(nchunks)
    cap = (L.pco_gfx_guarantee_chunk_size(nums.size, dtb) + 64 + 15) // 16 * 16
    dst = torch.zeros(nchunks * cap, dtype=torch.uint8, device="cuda")
    tasks = (G.EncodeTask * nchunks)()
    for i in range(nchunks):
        tasks[i] = G.EncodeTask(src.data_ptr() + i * nums.nbytes, nums.size, dst.data_ptr() + i * cap, cap, dtb, 0)
    res = (G.TaskResult * nchunks)()
    torch.cuda.synchronize()
    for it in range(3):
        t0 = time.time(); code = L.pco_gfx_compress_chunks(nchunks, tasks, C.byref(gcfg), res, None, None); torch.cuda.synchronize(); t1 = time.time()
        G.check(code)
        print(f"batched encode {kind} x{nchunks}: {t1-t0:.4f}s {nchunks*nums.nbytes/(t1-t0)/1e9:.1f} GB/s", flush=True)
    want = O.simple_compress(nums, ocfg)
    hdr = len(want) - 1 - res[0].n_out
    got0 = bytes(dst[: res[0].n_out].cpu().numpy()); gotl = bytes(dst[(nchunks - 1) * cap: (nchunks - 1) * cap + res[nchunks - 1].n_out].cpu().numpy())
    okb = got0 == want[hdr:-1] and gotl == want[hdr:-1]
    print("  chunk bytes == oracle:", okb, flush=True); ok_all &= okb
    # decode what we encoded
    out = torch.empty(nchunks * nums.size, dtype=src.dtype, device="cuda")
    dt = (G.DecodeTask * nchunks)()
    for i in range(nchunks):
        dt[i] = G.DecodeTask(dst.data_ptr() + i * cap, res[i].n_out, out.data_ptr() + i * nums.nbytes, nums.size, dtb, 0)
    dres = (G.TaskResult * nchunks)()
    for it in range(3):
        t0 = time.time(); code = L.pco_gfx_decompress_chunks(nchunks, dt, dres, None, None); torch.cuda.synchronize(); t1 = time.time()
        G.check(code)
        print(f"batched decode {kind} x{nchunks}: {t1-t0:.4f}s {nchunks*nums.nbytes/(t1-t0)/1e9:.1f} GB/s", flush=True)
    okd = bool(torch.equal(out, src)); print("  roundtrip == input:", okd, flush=True); ok_all &= okd
print("ALL OK" if ok_all else "SOME FAILED")
