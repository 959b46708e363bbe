"""Which chunks does the publishing walker mark for the expanders under the walk?  (noisy u64 ramps of several sizes, 2048 copies each)"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gpu_util as U, oracle_lib as O
from pcodec_amd import _lib as G
import test_gpu_parity as T
L = G.lib(); rng = np.random.default_rng(77)
for n in (700, 1000, 2048, 4096, 70000):
    a = (np.arange(n, dtype=np.int64) * 1000 + rng.integers(0, 512, n) + (1 << 30)).astype(np.uint64)
    f = O.simple_compress(a, O.make_config(mode=1, delta=2, delta_order=1)); b = U.chunk_of_file(f, len(f) - T.O_header_len(f) - 1)
    info, bins = O.inspect_first_chunk(f)
    k = 2048
    src = torch.from_numpy(np.frombuffer(b + b"\0" * 16, np.uint8).copy()).cuda()
    out = torch.zeros((k, a.nbytes), dtype=torch.uint8, device="cuda")
    dt = (G.DecodeTask * k)(*[G.DecodeTask(src.data_ptr(), len(b), out[i].data_ptr(), a.size, 2, 0) for i in range(k)])
    dr = (G.TaskResult * k)()
    m0, g0 = L.pco_gfx_trail_marked(), L.pco_gfx_trail_givebacks()
    G.check(L.pco_gfx_decompress_chunks(k, dt, dr, None, None))
    ok = bool((out.cpu().numpy() == a.view(np.uint8).reshape(1, -1)).all())
    print(f"n={n}: n_bins {info.n_bins[1]} ans_size_log {info.ans_size_log[1]} max offset bits {int(np.asarray(bins[1])[:, 2].max()) if len(bins[1]) else 0}; marked {L.pco_gfx_trail_marked() - m0} of {k}, given back {L.pco_gfx_trail_givebacks() - g0}, exact {ok}", flush=True)
