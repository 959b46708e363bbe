"""PCIe-inclusive rate of the host-buffer entry points (pco_standalone_simple_*): numbers for DESIGN.md section 6."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import pcodec_amd as P

cfg = P.ChunkConfig(mode_spec=P.ModeSpec.classic(), delta_spec=P.DeltaSpec.try_consecutive(1))
rng = np.random.default_rng(2)
for n in (1 << 18, 1 << 24):
    nums = (np.uint64(1 << 40) + np.uint64(1000) * np.arange(n, dtype=np.uint64) + rng.integers(0, 512, n).astype(np.uint64))
    blob = P.standalone.simple_compress(nums, cfg)
    out = P.standalone.simple_decompress(blob)
    assert np.array_equal(out, nums)
    reps = 20 if n <= (1 << 18) else 5
    t0 = time.perf_counter()
    for _ in range(reps): blob = P.standalone.simple_compress(nums, cfg)
    t1 = time.perf_counter()
    for _ in range(reps): out = P.standalone.simple_decompress(blob)
    t2 = time.perf_counter()
    gb = nums.nbytes / 1e9
    print(f"n = 2^{int(np.log2(n))} u64 ({nums.nbytes >> 20} MiB, {len(blob) >> 10} KiB compressed): simple_compress {(t1 - t0) / reps * 1e3:.2f} ms = {gb * reps / (t1 - t0):.2f} GB/s,"
          f" simple_decompress {(t2 - t1) / reps * 1e3:.2f} ms = {gb * reps / (t2 - t1):.2f} GB/s (host buffers, PCIe both ways, one call at a time)")
