"""GPU encode == oracle bytes on the reference's synthetic distributions (pco_cli/generate_randoms.py, as restated in
scripts/hist_fallback_census.py) at the chunk size 2^18: every histogram kernel and both walk arrangements see every shape.
usage: python scripts/dist_sweep_gpu.py [seeds]     (needs an MI355X)"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import oracle_lib as O, gpu_util as U
from pcodec_amd import _lib as G
src = open(os.path.join(ROOT, "scripts", "hist_fallback_census.py")).read()
ns = {"np": np}
exec(src[src.index("def lomax("):src.index("NP = {")], ns)   # the generators only
GENS = ns["GENS"]
NP = {"i64": np.int64, "u64": np.uint64, "i32": np.int32, "u32": np.uint32, "i16": np.int16, "u8": np.uint8, "f64": np.float64, "f32": np.float32, "f16": np.float16}
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = 1 << 18
CFGS = {"classic_nodelta": dict(mode=1, delta=1), "classic_delta1": dict(mode=1, delta=2, delta_order=1), "auto": dict()}
bad = 0; total = 0; differ_fb = 0; t0 = time.time()
for name, (fn, dts) in GENS.items():
    for dt in dts:
        if dt in ("u8",): continue
        for seed in range(seeds):
            x = np.asarray(fn(np.random.default_rng(1000 + seed), N))
            x = x.astype(NP[dt]) if NP[dt] != x.dtype else x
            for cname, kw in CFGS.items():
                ocfg = O.make_config(**kw)
                want = O.simple_compress(x, ocfg)
                got = U.gpu_simple_compress(x, G.make_config(**kw))
                total += 1
                if got != want:
                    _, _, fb = O.chunk_plan(x, ocfg)
                    if fb and U.bits_equal(O.simple_decompress(got, x.dtype, cap=N + 8), x): differ_fb += 1   # the reference's order-dependent heapsort branch
                    else: bad += 1; print("MISMATCH", name, dt, cname, seed, len(got), len(want), flush=True)
                elif not U.bits_equal(U.gpu_simple_decompress(got, x.dtype, N), x): bad += 1; print("DECODE MISMATCH", name, dt, cname, seed, flush=True)
print(f"{total} chunk encodes compared, {bad} mismatches, {differ_fb} differing only where the reference's heapsort fallback ran, {time.time() - t0:.0f} s")
