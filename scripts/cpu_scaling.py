"""Thread scaling of the oracle's native bench driver on this box (diagnostic for bench.py's cpu_baseline)."""
import ctypes as C, os, sys, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, oracle_lib as O, gpu_util as U
print("nproc", subprocess.run(["nproc"], capture_output=True, text=True).stdout.strip(), "cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
L = O.lib(); x = U.synth("c2"); cfg = O.make_config(mode=1, delta=2, delta_order=1)
for nt in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    if nt > 2 * (os.cpu_count() or 1): break
    out = (C.c_double * 4)()
    rc = L.pco_oracle_bench(x.ctypes.data_as(C.c_void_p), C.c_size_t(x.size), C.c_uint8(O.dtype_byte(x)), C.byref(cfg), C.c_uint32(nt), C.c_double(2.0), out)
    done, wall, se, sd = list(out)
    print(f"threads {nt:4d} rc {rc} chunks {int(done):6d} both {2 * done * x.nbytes / wall / 1e9:8.2f} GB/s  per-thread enc {done * x.nbytes / se / 1e9:6.3f} dec {done * x.nbytes / sd / 1e9:6.3f}", flush=True)
