#!/usr/bin/env python3
"""bench.py -- whole-job encode+decode throughput of the pcodec chunk hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic chunks that are already resident in HBM:
pco_gfx_compress_chunks over every chunk of the rank (ONE call), then pco_gfx_decompress_chunks over the chunks just
produced (ONE call).  Default workload = BASELINE.json configs[1]: u64, classic mode, delta order 1, 2^18-element
chunks of a noisy linear ramp.  value = uncompressed GB/s over encode+decode, 2*bytes/(t_enc+t_dec), aggregated over all
ranks (chunks are independent: each rank owns a contiguous block of chunks, weak scaling, no collective on the data
path).  `--gather` adds the file-assembly path of BASELINE configs[4]: device-side compaction of the rank's chunks,
gather-v of the exact byte ranges to rank 0 over RCCL, and for the decode direction the scatter of the byte ranges back --
through the library's own C ABI (pco_gfx_gather_chunks / pco_gfx_scatter_chunks, what INTEGRATION.md's Rust shim binds;
`--gather-carrier torch` for the torch.distributed carrier).  The default invocation times that leg too (workload `c5gather`).

Output: ONE JSON line on stdout, kept compact and FLAT where it matters (the driver's record keeps scalars two levels deep
and a ~10 KB tail): roofline.frac_encode / frac_decode / frac_step, roofline.traffic_over_algorithmic_*, roofline.ms.<kernel>,
and per extra workload config.<name>_value / _encode_GBps / _decode_GBps / _frac_step / ...  The full nested record (per-kernel
times and direction rooflines of every workload) goes to stderr as one line prefixed "BENCH_FULL " and to bench_full.json.

With --gpus N > 1 and no torch.distributed environment the script launches the N ranks itself
(torch.distributed.run, one rank per GPU, RCCL); under an external launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.
PyTorch is only plumbing here (device buffers, streams, process group); the codec is libpco_gfx.so.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N18 = 1 << 18
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

PAGE_INFO = np.dtype([("offset", "<u8"), ("len", "<u8"), ("n", "<u8"), ("status", "<u4"), ("aux", "<u4")])
PAGE_TASK = np.dtype([("meta", "<u8"), ("meta_len", "<u8"), ("page", "<u8"), ("page_len", "<u8"), ("dst", "<u8"), ("page_n", "<u8"), ("dtype", "<u4"), ("format_major", "<u4")])
ENC_TASK = np.dtype([("src", "<u8"), ("n", "<u8"), ("dst", "<u8"), ("dst_cap", "<u8"), ("dtype", "<u4"), ("reserved", "<u4")])
DEC_TASK = np.dtype([("src", "<u8"), ("src_len", "<u8"), ("dst", "<u8"), ("dst_cap", "<u8"), ("dtype", "<u4"), ("flags", "<u4")])
RESULT = np.dtype([("n_out", "<u8"), ("consumed", "<u8"), ("status", "<u4"), ("aux", "<u4")])

# chunk kinds: (numpy dtype, pco dtype byte)  -- SURVEY.md section 8(d) synthetic inputs
KINDS = {
    "u64ramp": (np.uint64, 2),      # C2: x[i] = 2^40 + 1000 i + U[0, 512)
    "f64cents": (np.float64, 6),    # C3: U{1000..9999} / 100 (pco_cli/generate_randoms.py:283-286)
    "u32rand": (np.uint32, 1),      # C1: uniform random 32-bit
    "i64season": (np.int64, 4),     # C4: period-365 seasonal + U{-3..3}
    "f32normal": (np.float32, 5),   # C5: standard normal (generate_randoms.py:242-244)
    "i32lomax": (np.int32, 3),      # C5: lomax(0.5) * 10 (generate_randoms.py:167-169)
}
WORKLOADS = {
    # name: (chunk kinds cycled over the chunk index, config kwargs, dtype label, description)
    "c2": (["u64ramp"], dict(mode=1, delta=2, delta_order=1), "u64", "u64 classic delta-1 noisy ramp, 2^18-element chunks (BASELINE configs[1])"),
    "c3": (["f64cents"], dict(mode=2, mode_f64=0.01, delta=1), "f64", "f64 float-mult(0.01) decimals, 2^18-element chunks (BASELINE configs[2])"),
    "c1": (["u32rand"], dict(mode=1, delta=1), "u32", "u32 classic no-delta uniform random, 2^18-element chunks (BASELINE configs[0], incompressible)"),
    "c4": (["i64season"], dict(mode=1, delta=3), "i64", "i64 seasonal (period 365) lookback delta, 2^18-element chunks (BASELINE configs[3])"),
    "c2auto": (["u64ramp"], dict(), "u64", "u64 noisy ramp, default ChunkConfig (Auto mode + Auto delta), 2^18-element chunks"),
    "c3auto": (["f64cents"], dict(), "f64", "f64 decimals, default ChunkConfig (Auto mode + Auto delta), 2^18-element chunks (SURVEY 8d C3, Auto)"),
    "c5": (["u64ramp", "f32normal", "i32lomax"], dict(mode=1, delta=2, delta_order=1), "u64/f32/i32",
           "mixed u64 ramp / f32 normal / i32 lomax chunks of 2^18, one call per rank, Classic + TryConsecutive(1) (BASELINE configs[4], explicit specs)"),
    # the headline's data with PCO_GFX_CFG_STRICT_HISTOGRAM (the reference's quickselect histogram replayed pivot by pivot: what bit-identity on
    # adversarial input ORDERS costs) and at compression level 12 (a published data point of the reference: 4096 histogram bins per chunk)
    "c2strict": (["u64ramp"], dict(mode=1, delta=2, delta_order=1, strict_histogram=True), "u64", "u64 classic delta-1 noisy ramp, strict (literal) histograms, 2^18-element chunks"),
    "c2l12": (["u64ramp"], dict(mode=1, delta=2, delta_order=1, level=12), "u64", "u64 classic delta-1 noisy ramp, compression level 12, 2^18-element chunks"),
    # the headline's data through the WRAPPED surface, batched (include/pco_gfx.h section 4b): 2^18-number chunks cut into pages of 16384 numbers
    # (PagingSpec::EqualPagesUpTo(16384): sixteen independent tANS streams per chunk), pco_gfx_compress_wrapped_chunks + pco_gfx_decompress_pages --
    # what the reference's `pcopage` bench codec does chunk by chunk (pco_cli/src/bench/codecs/pcopage.rs:33-113)
    "c2paged": (["u64ramp"], dict(mode=1, delta=2, delta_order=1, max_page_n=16384), "u64", "u64 classic delta-1 noisy ramp, 2^18-element chunks in pages of 16384 (wrapped surface, batched)"),
    "c5auto": (["u64ramp", "f32normal", "i32lomax"], dict(), "u64/f32/i32",
               "mixed u64 ramp / f32 normal / i32 lomax chunks of 2^18, one call per rank, default ChunkConfig (BASELINE configs[4], Auto/Auto)"),
}
MODE_NAMES = {0: "Auto", 1: "Classic", 2: "TryFloatMult", 3: "TryFloatQuant", 4: "TryIntMult"}
DELTA_NAMES = {0: "Auto", 1: "NoOp", 2: "TryConsecutive", 3: "TryLookback"}


def make_kind(torch, kind, n_chunks, g, device):
    """[n_chunks, 2^18] device tensor of one chunk kind (torch dtype with the same bits as the pco dtype)."""
    if kind == "u64ramp":
        i = torch.arange(N18, device=device, dtype=torch.int64)
        noise = torch.randint(0, 512, (n_chunks, N18), generator=g, device=device, dtype=torch.int64)
        start = torch.randint(0, 1 << 20, (n_chunks, 1), generator=g, device=device, dtype=torch.int64)
        return ((1 << 40) + 1000 * i).unsqueeze(0) + noise + start   # non-negative: same bits as u64
    if kind == "f64cents":
        return torch.randint(1000, 10000, (n_chunks, N18), generator=g, device=device, dtype=torch.int64).to(torch.float64) / 100.0
    if kind == "u32rand":
        return torch.randint(-(1 << 31), 1 << 31, (n_chunks, N18), generator=g, device=device, dtype=torch.int64).to(torch.int32)
    if kind == "i64season":
        base = torch.randint(-(1 << 40), 1 << 40, (365,), generator=g, device=device, dtype=torch.int64)
        idx = torch.arange(N18, device=device) % 365
        return base[idx].unsqueeze(0) + torch.randint(-3, 4, (n_chunks, N18), generator=g, device=device, dtype=torch.int64)
    if kind == "f32normal":
        return torch.randn((n_chunks, N18), generator=g, device=device, dtype=torch.float32)
    if kind == "i32lomax":   # numpy's pareto(a) is Lomax: (1 - U)^(-1/a) - 1, a = 0.5
        u = torch.rand((n_chunks, N18), generator=g, device=device, dtype=torch.float64)
        return (((1.0 - u) ** -2.0 - 1.0) * 10.0).clamp(0, 2e9).to(torch.int32)
    raise KeyError(kind)


def host_sample(kind, seed=0):
    """One chunk of a kind on the host (numpy) for the CPU baseline."""
    import gpu_util as U
    rng = np.random.default_rng(1000 + seed)
    if kind == "u64ramp": return U.synth("c2")
    if kind == "f64cents": return U.synth("c3")
    if kind == "u32rand": return U.synth("c1")
    if kind == "i64season": return U.synth("c4")
    if kind == "f32normal": return rng.standard_normal(N18).astype(np.float32)
    if kind == "i32lomax": return (rng.pareto(0.5, N18) * 10).clip(0, 2e9).astype(np.int32)
    raise KeyError(kind)


def cpu_info():
    info = {"logical_cpus": os.cpu_count() or 1}
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        for line in out.splitlines():
            k, _, v = line.partition(":"); k = k.strip(); v = v.strip()
            if k == "Model name": info["model"] = v
            elif k == "Socket(s)": info["sockets"] = int(v)
            elif k == "Core(s) per socket": info["cores_per_socket"] = int(v)
            elif k == "Thread(s) per core": info["threads_per_core"] = int(v)
    except Exception:
        pass
    if "sockets" in info and "cores_per_socket" in info:
        info["physical_cores"] = info["sockets"] * info["cores_per_socket"]
    # a container may own fewer CPUs than it sees: cgroup v2 cpu.max = "<quota> <period>" (v1: cpu.cfs_quota_us / cpu.cfs_period_us)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max": info["cgroup_cpu_quota"] = round(int(q) / int(p), 2)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0: info["cgroup_cpu_quota"] = round(q / p, 2)
        except Exception:
            pass
    try:
        info["affinity_cpus"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    return info


def oracle_kw(cfg_kw):
    """The reference's ChunkConfig has no strict flag: its histogram IS the literal one."""
    return {k: v for k, v in cfg_kw.items() if k != "strict_histogram"}


def cpu_baseline(kinds, cfg_kw, seconds=14.0):
    """The oracle (a C++ restatement of the reference's algorithm, NOT the Rust binary) on the host cores: a native
    std::thread driver inside oracle/ (pco_oracle_bench) -- every thread loops compress -> decompress over a private copy
    of one chunk, no Python in the loop.  One thread first (per-direction rates), then one thread per logical CPU."""
    import oracle_lib as O

    L = O.lib()
    ocfg = O.make_config(**oracle_kw(cfg_kw))
    info = cpu_info()
    # one thread per CPU this process may actually use: more threads than the cgroup quota only get throttled (measured on the
    # round-2 GPU box: cpu.max = 16 CPUs of a 2 x 64-core EPYC 9575F; linear to 16 threads, 13.7 GB/s, then falling)
    threads = int(min(info["logical_cpus"], info.get("affinity_cpus", 1 << 30), max(1, int(info.get("cgroup_cpu_quota", 1 << 30)))))
    per_kind = seconds / len(kinds)
    one = {"chunks": 0, "bytes": 0.0, "enc_s": 0.0, "dec_s": 0.0}
    many = {"chunks": 0, "bytes": 0.0, "wall": 0.0}
    for k, kind in enumerate(kinds):
        x = host_sample(kind, k)
        out = (C.c_double * 4)()

        def run(nt, secs):
            rc = L.pco_oracle_bench(x.ctypes.data_as(C.c_void_p), C.c_size_t(x.size), C.c_uint8(O.dtype_byte(x)), C.byref(ocfg), C.c_uint32(nt), C.c_double(secs), out)
            if rc != 0:
                raise RuntimeError("oracle bench failed: " + L.pco_oracle_last_error().decode())
            return list(out)

        done, wall, se, sd = run(1, per_kind * 0.25)
        one["chunks"] += done; one["bytes"] += done * x.nbytes; one["enc_s"] += se; one["dec_s"] += sd
        done, wall, se, sd = run(threads, per_kind * 0.75)
        many["chunks"] += done; many["bytes"] += done * x.nbytes; many["wall"] += wall
    single = {"enc_gbs": one["bytes"] / one["enc_s"] / 1e9, "dec_gbs": one["bytes"] / one["dec_s"] / 1e9, "both_gbs": 2 * one["bytes"] / (one["enc_s"] + one["dec_s"]) / 1e9}
    value = 2 * many["bytes"] / many["wall"] / 1e9
    return {"value": round(value, 3), "unit": "GB/s", "cores": threads, "kind": "port",
            "sample": f"{int(many['chunks'])} chunks of 2^18 ({'/'.join(kinds)}) encode+decode over {threads} native threads in {many['wall']:.1f}s, after "
                      f"{int(one['chunks'])} chunks on one thread (C++ restatement of the reference, g++ -O3 -mavx2, std::thread driver in oracle/; "
                      "the Rust reference cannot be built here)",
            "single_thread": {k: round(v, 3) for k, v in single.items()},
            "scaling_vs_single_thread": round(value / single["both_gbs"], 1),
            # what the same code would reach on all physical cores of ONE socket if it kept scaling linearly (it does up to the quota);
            # an upper bound for the "single-socket CPU" the north star compares with, not a measurement
            "linear_extrapolation_one_socket": round(single["both_gbs"] * info["cores_per_socket"], 1) if "cores_per_socket" in info else None,
            "cpu": info}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: run the N ranks under torch.distributed.run (RCCL) and mirror rank 0's line."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def csrc_sha16():
    """Hash of the product's kernel sources: a committed PMC traffic table is only quoted for the build it was measured on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pcodec_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if os.path.isfile(os.path.join(d, f)) and f.endswith((".hip", ".inc", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def load_traffic(workload, nch, kernel_names):
    """HBM bytes per kernel from the committed PMC passes (profiles/rNN_traffic_<workload>.json), scaled to this run's chunk count --
    or (None, why) when there is no table for this workload, when it was measured on other kernel sources (csrc_sha16), or when the
    kernels it lists are not the kernels that just ran."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_traffic_{workload}.json")), reverse=True)
    if not cands:
        return None, "no PMC table committed for this workload"
    try:
        tj = json.load(open(cands[0]))
        if tj.get("workload") != workload:
            return None, "table names another workload"
        if tj.get("csrc_sha16") != csrc_sha16():
            return None, f"{os.path.basename(cands[0])} was measured on other kernel sources (csrc_sha16 {tj.get('csrc_sha16')}); re-run scripts/pmc_run.sh"
        tab = {k: int((v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * nch / tj["chunks"]) for k, v in tj["kernels"].items()}
        kernel_names = [k.lstrip("~") for k in kernel_names if "+" not in k]   # (spans of concurrent kernels are not kernels)
        missing = [k for k in kernel_names if k not in tab and not k.startswith(("pco_decode_kernel", "enc_page_kernel", "enc_init", "enc_presample", "enc_scan", "enc_lat_slots", "dec_walk4", "dec_walk_kernel(rest)"))]
        if missing:
            return None, f"{os.path.basename(cands[0])} lacks kernels that ran: {missing}"
        return tab, os.path.basename(cands[0])
    except (OSError, ValueError, KeyError) as e:
        return None, f"unreadable table: {e}"


class Bench:
    def __init__(self, args):
        import torch
        import torch.distributed as dist
        from pcodec_amd import _lib as G
        from pcodec_amd import sharding as S
        self.torch, self.dist, self.G, self.S, self.args = torch, dist, G, S, args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device: pcodec_amd has no CPU fallback")
        if self.world != args.gpus and self.rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}; reporting n_gpus={self.world}", file=sys.stderr)
        torch.cuda.set_device(local_rank)
        self.device = torch.device("cuda", local_rank)
        # PCO_BENCH_FORCE_DIST=1: go through the process group (RCCL init, barrier, all-reduce, gather-v) at world size 1 too, so that a
        # single-GPU box exercises the code an 8-GPU run depends on
        self.use_dist = self.world > 1 or os.environ.get("PCO_BENCH_FORCE_DIST") == "1"
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
            dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.device)
        self.comm = None
        self.L = G.lib()
        self.L.pco_gfx_compact_chunks.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]

    def c_abi_comm(self):
        """The library's own communicator (include/pco_gfx.h section 5): rank 0 makes the 128-byte id, the process group (the host's
        channel here) carries it to the others; at world size 1 nothing else is needed -- RCCL still runs (ncclCommInitRank, the size
        all-gather)."""
        if self.comm is None:
            ident = [self.S.Comm.unique_id() if self.rank == 0 else None]
            if self.world > 1:
                self.dist.broadcast_object_list(ident, src=0)
            self.comm = self.S.Comm(ident[0], self.world, self.rank)
        return self.comm

    def sync_all(self):
        self.torch.cuda.synchronize()
        if self.use_dist:
            self.dist.barrier()

    def verify_against_oracle(self, kinds, cfg_kw, kind_of, row_of, data, comp, cap_off, n_out, k_verify):
        """Rank 0, warm-up: `k_verify` randomly drawn chunks, compressed by the oracle on every host CPU this process owns (native
        threads: oracle/pco_oracle_capi.cpp pco_oracle_verify_chunks) and compared byte for byte with what the GPU wrote."""
        import oracle_lib as O
        torch = self.torch
        nch = len(kind_of)
        pick = np.sort(np.random.default_rng(99).choice(nch, size=min(k_verify, nch), replace=False))
        info = cpu_info()
        threads = int(min(info["logical_cpus"], info.get("affinity_cpus", 1 << 30), max(1, int(info.get("cgroup_cpu_quota", 1 << 30)))))
        OL = O.lib(); ocfg = O.make_config(**oracle_kw(cfg_kw))
        checked = 0
        for k, kind in enumerate(kinds):
            idx = pick[kind_of[pick] == k]
            if len(idx) == 0:
                continue
            rows = torch.from_numpy(row_of[idx]).to(self.device)
            host = data[k][rows].cpu().numpy()   # [len(idx), 2^18] of the kind's torch dtype: same bits as the pco dtype
            offs = np.zeros(len(idx), np.uint64); lens = n_out[idx].astype(np.uint64)
            offs[1:] = np.cumsum(lens[:-1])
            got = torch.empty(int(lens.sum()) + 64, dtype=torch.uint8, device=self.device)
            # (gather the picked chunks' bytes on the device, one copy back)
            for j, i in enumerate(idx):
                got[int(offs[j]): int(offs[j]) + int(lens[j])] = comp[int(cap_off[i]): int(cap_off[i]) + int(lens[j])]
            got_h = got.cpu().numpy()
            n_bad = C.c_uint64(0); first = C.c_int64(-1)
            rc = OL.pco_oracle_verify_chunks(host.ctypes.data_as(C.c_void_p), C.c_size_t(len(idx)), C.c_size_t(N18), C.c_size_t(host.strides[0]), C.c_uint8(KINDS[kind][1]),
                                             C.byref(ocfg), got_h.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p),
                                             C.c_uint32(threads), C.byref(n_bad), C.byref(first))
            if rc != 0:
                raise RuntimeError("oracle verify failed: " + OL.pco_oracle_last_error().decode())
            assert n_bad.value == 0, f"{n_bad.value} GPU chunks of kind {kind} differ from the oracle's bytes (first: chunk {int(idx[first.value])})"
            checked += len(idx)
        return checked

    def run(self, workload, chunks, steps, warmup, gather, verify_chunks, with_cpu, cpu_seconds=14.0, carrier="cabi"):
        """One workload: data generation, warm-up (round-trip assertion + oracle spot check), K timed steps.  Returns rank 0's record."""
        torch, dist, G, S, L = self.torch, self.dist, self.G, self.S, self.L
        world, rank, device = self.world, self.rank, self.device
        kinds, cfg_kw, dtype_label, desc = WORKLOADS[workload]
        gcfg = G.make_config(**cfg_kw)
        elem = [np.dtype(KINDS[k][0]).itemsize for k in kinds]
        if self.args.total_gib and chunks is None:   # STRONG scaling: `--total-gib G` of numbers over ALL ranks (BASELINE configs[4]: "~64 GiB total" at 1/2/4/8 GPUs)
            per_cycle = sum(N18 * e for e in elem)
            total_chunks = max(len(kinds) * world, int((self.args.total_gib * (1 << 30)) // per_cycle) * len(kinds))
            chunks = -(-total_chunks // (world * len(kinds))) * len(kinds)
        if chunks is None:   # 16 GiB of numbers per GPU
            per_cycle = sum(N18 * e for e in elem)
            chunks = max(len(kinds), int((16 << 30) // per_cycle) * len(kinds))
        nch = -(-chunks // len(kinds)) * len(kinds)   # whole cycles of the kinds: every rank owns the same mix
        # this rank's block of the global chunk sequence (contiguous blocks, sharding.shard_range); kinds cycle over the GLOBAL index
        c0, c1 = S.shard_range(nch * world, rank, world)
        assert c1 - c0 == nch
        g = torch.Generator(device=device); g.manual_seed(1234 + 7919 * rank)
        kind_of = np.array([(c0 + i) % len(kinds) for i in range(nch)])
        data = {}; row_of = np.zeros(nch, np.int64)
        for k, kind in enumerate(kinds):
            idx = np.nonzero(kind_of == k)[0]
            row_of[idx] = np.arange(len(idx))
            if len(idx): data[k] = make_kind(torch, kind, len(idx), g, device).contiguous()
        out = {k: torch.empty_like(v) for k, v in data.items()}
        chunk_bytes = np.array([N18 * elem[k] for k in kind_of], dtype=np.uint64)
        dtb = np.array([KINDS[kinds[k]][1] for k in kind_of], dtype=np.uint32)
        paged = "max_page_n" in cfg_kw
        if paged:   # room for the ChunkMeta and every page of a chunk (pco_gfx_wrapped_chunk_cap)
            L.pco_gfx_wrapped_chunk_cap.restype = C.c_size_t; L.pco_gfx_wrapped_chunk_cap.argtypes = [C.c_size_t, C.c_ubyte, C.c_void_p]
            L.pco_gfx_wrapped_n_pages.restype = C.c_size_t; L.pco_gfx_wrapped_n_pages.argtypes = [C.c_size_t, C.c_uint64]
            caps = np.array([(L.pco_gfx_wrapped_chunk_cap(N18, int(b), C.addressof(gcfg)) + 64 + 15) // 16 * 16 for b in dtb], dtype=np.uint64)
        else:
            caps = np.array([(L.pco_gfx_guarantee_chunk_size(N18, int(b)) + 64 + 15) // 16 * 16 for b in dtb], dtype=np.uint64)
        cap_off = np.concatenate([[0], np.cumsum(caps)]).astype(np.uint64)
        comp = torch.zeros(int(cap_off[-1]), dtype=torch.uint8, device=device)
        src_ptr = np.array([data[k].data_ptr() + int(r) * N18 * elem[k] for k, r in zip(kind_of, row_of)], dtype=np.uint64)
        out_ptr = np.array([out[k].data_ptr() + int(r) * N18 * elem[k] for k, r in zip(kind_of, row_of)], dtype=np.uint64)

        enc_tasks = np.zeros(nch, ENC_TASK)
        enc_tasks["src"] = src_ptr; enc_tasks["n"] = N18; enc_tasks["dtype"] = dtb; enc_tasks["dst_cap"] = caps
        enc_tasks["dst"] = np.uint64(comp.data_ptr()) + cap_off[:-1]
        dec_tasks = np.zeros(nch, DEC_TASK)
        dec_tasks["src"] = enc_tasks["dst"]; dec_tasks["dst"] = out_ptr; dec_tasks["dst_cap"] = N18; dec_tasks["dtype"] = dtb
        enc_res = np.zeros(nch, RESULT); dec_res = np.zeros(nch, RESULT)
        d_res = torch.zeros(nch * RESULT.itemsize, dtype=torch.uint8, device=device)
        # --gather buffers: this rank's compacted stream, its offsets, the scattered copy the decoders read, and (rank 0) the file body
        if gather:
            stream_cap = int(cap_off[-1]) + 64
            payload = torch.zeros(stream_cap, dtype=torch.uint8, device=device)
            d_offs = torch.zeros(nch + 1, dtype=torch.int64, device=device)
            recv = torch.zeros(stream_cap, dtype=torch.uint8, device=device)
            use_cabi = carrier == "cabi"
            moves = use_cabi or self.use_dist      # the torch carrier needs a process group; the C-ABI one brings its own communicator
            file_cap = stream_cap * world
            file_body = torch.zeros(file_cap if rank == 0 else 16, dtype=torch.uint8, device=device) if moves else None
            comm = self.c_abi_comm() if use_cabi else None
        gather_ms = []
        state = {"n_bytes": 0, "offs": None}
        if paged:
            assert len(kinds) == 1 and not gather
            npg = int(L.pco_gfx_wrapped_n_pages(N18, cfg_kw["max_page_n"]))
            infos = np.zeros((nch, 1 + npg), PAGE_INFO)
            ptasks = np.zeros((nch, npg), PAGE_TASK)
            pres = np.zeros(nch * npg, RESULT)
            ptasks["dtype"] = dtb[:, None]; ptasks["format_major"] = 4; ptasks["meta"] = enc_tasks["dst"][:, None]

        def encode_paged():
            G.check(L.pco_gfx_compress_wrapped_chunks(nch, enc_tasks.ctypes.data, C.addressof(gcfg), infos.ctypes.data, None))
            enc_res["n_out"] = infos["len"].sum(axis=1)   # a chunk's bytes: its ChunkMeta + its pages

        def decode_paged():   # one task per page: the chunk's ChunkMeta + the page's own bytes -> the page's slice of the chunk's numbers
            ptasks["meta_len"] = infos["len"][:, 0:1]
            ptasks["page"] = enc_tasks["dst"][:, None] + infos["offset"][:, 1:]
            ptasks["page_len"] = infos["len"][:, 1:]; ptasks["page_n"] = infos["n"][:, 1:]
            first = np.cumsum(infos["n"][:, 1:], axis=1) - infos["n"][:, 1:]
            ptasks["dst"] = out_ptr[:, None] + first * np.uint64(elem[0])
            G.check(L.pco_gfx_decompress_pages(nch * npg, ptasks.ctypes.data, pres.ctypes.data, None, None))

        def encode():
            if paged:
                return encode_paged()
            G.check(L.pco_gfx_compress_chunks(nch, enc_tasks.ctypes.data, C.byref(gcfg), enc_res.ctypes.data, d_res.data_ptr() if gather else None, None))
            if gather:   # device-side compaction (no Python loop), then the exact-size gather-v to rank 0
                t = time.perf_counter()
                total = C.c_uint64(0)
                G.check(L.pco_gfx_compact_chunks(nch, enc_tasks.ctypes.data, d_res.data_ptr(), payload.data_ptr(), stream_cap - 64, 0, d_offs.data_ptr(), C.byref(total), None))
                state["n_bytes"] = int(total.value)
                if use_cabi:
                    state["offs"] = comm.gather(payload.data_ptr(), state["n_bytes"], file_body.data_ptr(), file_cap, 0, root=0)
                    torch.cuda.synchronize()
                elif self.use_dist:
                    totals = S.exchange_totals(state["n_bytes"], device)
                    _, state["offs"] = S.gather_stream(payload, state["n_bytes"], dst=0, out=file_body, totals=totals)
                    torch.cuda.synchronize()
                gather_ms.append((time.perf_counter() - t) * 1e3)

        def decode():
            if paged:
                return decode_paged()
            if gather:   # decoders read the byte ranges the root hands out
                if use_cabi:
                    comm.scatter(file_body.data_ptr(), state["offs"], recv.data_ptr(), stream_cap - 16, 0, root=0)
                    base = recv.data_ptr()
                elif self.use_dist:
                    S.scatter_stream(file_body, state["offs"], recv, src=0)
                    base = recv.data_ptr()
                else:
                    base = payload.data_ptr()
                sizes = enc_res["n_out"]
                dec_tasks["src"] = np.uint64(base) + np.concatenate([[0], np.cumsum(sizes[:-1])]).astype(np.uint64)
                dec_tasks["src_len"] = sizes
            else:
                dec_tasks["src_len"] = enc_res["n_out"]
            G.check(L.pco_gfx_decompress_chunks(nch, dec_tasks.ctypes.data, dec_res.ctypes.data, None, None))

        # warm-up, with the reference bench's bitwise round-trip assertion (pco_cli/src/bench/codecs/mod.rs:176-189)
        L.pco_gfx_profile_begin()   # (warm-up also fills the library's pool of timing events, so the timed steps create none)
        for w in range(max(warmup, 1)):
            encode(); decode()
        torch.cuda.synchronize()
        no_verify = os.environ.get("PCO_BENCH_NO_VERIFY") == "1"   # ablation builds of the library (scripts/ab_*.sh) decode garbage on purpose
        for k in data:
            assert no_verify or torch.equal(out[k].view(torch.uint8), data[k].view(torch.uint8)), "decode(encode(x)) != x"
        verified = 0
        if rank == 0 and verify_chunks > 0 and not no_verify and paged:   # the oracle's wrapped::ChunkCompressor on a few chunks: ChunkMeta and every page, byte for byte
            import oracle_lib as O
            ocfg = O.make_config(**oracle_kw(cfg_kw))
            for i in np.sort(np.random.default_rng(99).choice(nch, size=min(verify_chunks, 48, nch), replace=False)):
                host = data[0][int(row_of[i])].cpu().numpy().view(KINDS[kinds[0]][0])
                want_meta, want_pages, want_ns = O.wrapped_compress(host, ocfg)
                blob = comp[int(cap_off[i]): int(cap_off[i]) + int(caps[i])].cpu().numpy()
                got_meta = bytes(blob[: int(infos[i, 0]["len"])])
                got_pages = [bytes(blob[int(e["offset"]): int(e["offset"]) + int(e["len"])]) for e in infos[i, 1:]]
                assert got_meta == want_meta and got_pages == want_pages and [int(e["n"]) for e in infos[i, 1:]] == want_ns, f"wrapped chunk {i} differs from the oracle's bytes"
                verified += 1
        elif rank == 0 and verify_chunks > 0 and not no_verify:   # parity spot check against the oracle's bytes (native threads)
            verified = self.verify_against_oracle(kinds, cfg_kw, kind_of, row_of, data, comp, cap_off, enc_res["n_out"], verify_chunks)

        # timed region: exactly K steps, bracketed by barrier + synchronize
        self.sync_all()
        gather_ms.clear()
        L.pco_gfx_profile_begin()
        t_enc = t_dec = 0.0
        step_ms = []
        t0 = time.perf_counter()
        for _ in range(steps):
            a = time.perf_counter(); encode(); b = time.perf_counter(); decode(); c = time.perf_counter()
            t_enc += b - a; t_dec += c - b; step_ms.append((c - a) * 1e3)
        self.sync_all()
        elapsed = time.perf_counter() - t0
        names = C.create_string_buffer(1 << 18); ms = (C.c_float * 65536)()
        nk = L.pco_gfx_profile_end(names, len(names), ms, 65536)
        el = torch.tensor([elapsed, t_enc, t_dec], device=device, dtype=torch.float64)
        if self.use_dist:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed, t_enc, t_dec = (float(x) for x in el.tolist())
        ms_per_step = elapsed * 1e3 / steps
        rank_bytes = int(chunk_bytes.sum())
        total_bytes = world * rank_bytes   # (every rank owns the same mix: kinds cycle with a period that divides nch)
        value = 2 * total_bytes / (ms_per_step * 1e-3) / 1e9
        rec = None
        if rank == 0:
            raw = names.raw; kn = []; pos = 0
            for _ in range(nk):
                e = raw.index(b"\0", pos); kn.append(raw[pos:e].decode()); pos = e + 1
            per = {}
            for nm, t in zip(kn, ms[:nk]):
                per.setdefault(nm, []).append(float(t))
            kavg = {k: sum(v) / len(v) for k, v in per.items()}                 # per launch
            kstep = {k: sum(v) / steps for k, v in per.items()}                 # per step (a kernel may launch several times per step)
            comp_bytes = int(enc_res["n_out"].sum())
            # algorithmic bytes per launch: encode = n*sizeof(T) read + C written; decode = C read + n*sizeof(T) written
            alg = rank_bytes + comp_bytes
            # kernel labels: "~name" = a kernel that ran CONCURRENTLY with others on a second stream (the decode walker and the expanders under
            # it); "a+b<..>" = the span of such a group on the caller's stream.  Direction sums take the spans and the ordinary kernels (a sum
            # of overlapping kernels would count the same milliseconds twice); the dominant KERNEL is looked for among real kernels.
            real = {k.lstrip("~"): v for k, v in kavg.items() if "+" not in k}
            dom = max(real, key=real.get)
            kavg = {**kavg, **real}
            traffic_tab, traffic_src = load_traffic(workload, nch, list(kstep))

            def direction(prefixes):
                ks = [k for k in kstep if k.startswith(prefixes)]   # ("~..." kernels are inside a span that is counted)
                t = sum(kstep[k] for k in ks)
                members = [k.lstrip("~") for k in kstep if k.lstrip("~").startswith(prefixes) and "+" not in k]
                tr = sum(traffic_tab.get(k, 0) for k in members) if traffic_tab else None
                return {"kernel_ms": round(t, 4), "achieved": round(alg / (t * 1e-3) / 1e9, 1) if t > 0 else None,
                        "frac": round(alg / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if t > 0 else None, "traffic": tr,
                        "traffic_over_algorithmic": round(tr / alg, 2) if tr else None}

            enc_d = direction(("enc_", "gather_", "compact_", "auto_", "split_gather")); dec_d = direction(("dec_", "pco_decode"))
            both_t = enc_d["kernel_ms"] + dec_d["kernel_ms"]
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(alg / (kavg[dom] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(alg / (kavg[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic_tab.get(dom) if traffic_tab else None,
                    "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": alg, "avg_launch_ms": round(kavg[dom], 4),
                    # the honest numbers: a direction's algorithmic bytes over the SUM of its kernels' time (the dominant-kernel
                    # figure above credits one kernel with the whole direction's bytes)
                    "direction": {"encode": enc_d, "decode": dec_d,
                                  "step": {"kernel_ms": round(both_t, 4), "achieved": round(2 * alg / (both_t * 1e-3) / 1e9, 1), "frac": round(2 * alg / (both_t * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}},
                    "per_kernel_ms_per_step": {k: round(v, 4) for k, v in sorted(kstep.items())},
                    # the same, flat (the driver's record keeps scalars at this depth and drops nested objects): the direction-level
                    # fractions are what the pipeline achieves; `frac` above is the contract's dominant-kernel figure
                    "frac_encode": enc_d["frac"], "frac_decode": dec_d["frac"], "frac_step": round(2 * alg / (both_t * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "kernel_ms_encode": enc_d["kernel_ms"], "kernel_ms_decode": dec_d["kernel_ms"],
                    "traffic_encode": enc_d["traffic"], "traffic_decode": dec_d["traffic"],
                    "traffic_over_algorithmic_encode": enc_d["traffic_over_algorithmic"], "traffic_over_algorithmic_decode": dec_d["traffic_over_algorithmic"]}
            rec = {
                "workload": workload, "value": round(value, 2), "ms_per_step": round(ms_per_step, 3), "dtype": dtype_label, "steps": steps, "warmup": warmup,
                "config": {"workload": desc, "chunks_per_gpu": nch, "chunk_n": N18, "compression_level": cfg_kw.get("level", 8), **({"strict_histogram": True} if cfg_kw.get("strict_histogram") else {}),
                           "mode_spec": MODE_NAMES[cfg_kw.get("mode", 0)] + (f"({cfg_kw['mode_f64']})" if "mode_f64" in cfg_kw else ""),
                           "delta_spec": DELTA_NAMES[cfg_kw.get("delta", 0)] + (f"({cfg_kw['delta_order']})" if "delta_order" in cfg_kw else ""),
                           **({"paging_spec": f"EqualPagesUpTo({cfg_kw['max_page_n']})", "pages_per_chunk": npg, "api": "pco_gfx_compress_wrapped_chunks / pco_gfx_decompress_pages"} if paged else {}),
                           **({"total_GiB_all_ranks": round(world * rank_bytes / (1 << 30), 2)} if self.args.total_gib else {}),
                           "parallelism": f"chunk-sharded x{world}, contiguous chunk blocks" + ((f", device compaction + RCCL gather-v / scatter of the chunk bytes ({'C ABI pco_gfx_gather_chunks / _scatter_chunks' if carrier == 'cabi' else 'torch.distributed carrier'})" if gather else ", no data-path collective")),
                           "compressed_bytes_per_chunk": comp_bytes // nch,
                           "encode_GBps": round(total_bytes * steps / t_enc / 1e9, 2),
                           "decode_GBps": round(total_bytes * steps / t_dec / 1e9, 2),
                           "median_step_ms_rank0": round(float(np.median(step_ms)), 3),
                           # the library's device scratch after the timed steps (it grows to what the calls needed and stays) per byte of numbers
                           "workspace_bytes_per_input_byte": round(L.pco_gfx_workspace_bytes() / rank_bytes, 3),
                           "oracle_verified_chunks": verified, **({"UNVERIFIED_ablation_run": True} if no_verify else {})},
                "roofline": roof,
            }
            if gather:
                rec["config"]["gather_ms_per_step_rank0"] = round(float(np.mean(gather_ms)), 3) if gather_ms else None
                rec["config"]["stream_bytes_per_rank"] = state["n_bytes"]
            if with_cpu:   # rank 0's host cores, at every world size (the other ranks wait at the barrier below, idle)
                rec["cpu_baseline"] = cpu_baseline(kinds, cfg_kw, seconds=cpu_seconds)
                if world > 1:
                    rec["cpu_baseline"]["sample"] += f"; timed on rank 0 while the other {world - 1} ranks were idle"
        if with_cpu and self.use_dist:
            dist.barrier()
        # give the memory back before the next workload
        del data, out, comp, d_res
        if gather:
            del payload, d_offs, recv, file_body
        L.pco_gfx_release_workspace()
        torch.cuda.empty_cache()
        return rec


# what the default invocation times after the headline workload (BASELINE.json's other configs + the default ChunkConfig on the
# headline's and the f64 data + configs[4] WITH its compaction / gather / scatter leg): (name, workload, chunks per GPU (None = 16 GiB
# of numbers), steps, gather)
OTHER_WORKLOADS = [("c3", "c3", None, 4, False), ("c4", "c4", 4096, 3, False), ("c5", "c5", None, 3, False), ("c5gather", "c5", None, 3, True),
                   ("c2auto", "c2auto", None, 3, False), ("c3auto", "c3auto", None, 3, False), ("c1", "c1", None, 3, False),
                   ("c5auto", "c5auto", None, 2, False),   # configs[4] under the default ChunkConfig (Auto mode + Auto delta per chunk)
                   ("c2strict", "c2strict", 4096, 2, False), ("c2l12", "c2l12", 2048, 2, False),
                   # the headline's dependence on the call size (the walkers' flat cost: profiles/r04_c2_chunk_scaling.txt); 4096 chunks per GPU is
                   # what configs[4]'s "~64 GiB over 8 GPUs" comes to
                   ("c2_1k", "c2", 1024, 3, False), ("c2_4k", "c2", 4096, 3, False), ("c2_12k", "c2", 12288, 3, False),
                   # the same 1024 chunks through the wrapped surface in pages of 16384 numbers: sixteen independent tANS streams per chunk, the honest
                   # answer to the walkers' flat cost in small calls (compare c2_1k)
                   ("c2paged", "c2paged", 1024, 3, False)]
LIGHT_WORKLOADS = ("c5auto", "c2strict", "c2l12", "c2_1k", "c2_4k", "c2_12k", "c2paged")   # no CPU leg of their own (the headline's / c2's applies), fewer verified chunks
# The driver's record keeps the FIRST 24 keys of `config` and of `roofline`: one scalar pair per BASELINE config and per default-ChunkConfig
# variant goes there, everything else behind
CONFIG_FIRST = ["workload", "chunks_per_gpu", "encode_GBps", "decode_GBps", "compressed_bytes_per_chunk",
                "c3_value", "c3_frac_step", "c4_value", "c4_frac_step", "c4_x_socket_extrapolated", "c5_value", "c5_frac_step", "c5gather_value", "c5gather_frac_step",
                "c2auto_value", "c2auto_frac_step", "c3auto_value", "c3auto_frac_step", "c1_value", "c1_frac_step", "c5auto_value", "c5auto_frac_step", "c2_4k_value", "c2paged_value"]
ROOFLINE_FIRST = ["bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_encode", "frac_decode", "frac_step", "kernel_ms_encode", "kernel_ms_decode",
                  "traffic_over_algorithmic_encode", "traffic_over_algorithmic_decode", "avg_launch_ms", "algorithmic_bytes_per_launch",
                  "ms.enc_split_kernel<c16>", "ms.enc_hist_kernel", "ms.enc_train_kernel", "ms.enc_walkp_kernel", "ms.enc_place+pack", "ms.enc_place_kernel", "ms.~enc_place_kernel", "ms.dec_walk+trail<u64>",
                  "traffic_source"]


def ordered(d, first):
    """`d` with the keys of `first` (those present) in front, the rest in their own order."""
    out = {k: d[k] for k in first if k in d}
    out.update({k: v for k, v in d.items() if k not in out})
    return out


def flat_workload(name, r):
    """The scalars of one extra workload that must survive in the driver's record: config.<name>_<key>."""
    d = r["roofline"]
    out = {"value": r["value"], "ms_per_step": r["ms_per_step"], "chunks": r["config"]["chunks_per_gpu"], "steps": r["steps"], "workspace_x": r["config"]["workspace_bytes_per_input_byte"],
           "encode_GBps": r["config"]["encode_GBps"], "decode_GBps": r["config"]["decode_GBps"],
           "frac_encode": d["frac_encode"], "frac_decode": d["frac_decode"], "frac_step": d["frac_step"],
           "traffic_x_encode": d["traffic_over_algorithmic_encode"], "traffic_x_decode": d["traffic_over_algorithmic_decode"],
           "top_kernel": f"{d['kernel']} {d['avg_launch_ms']} ms", "verified": r["config"]["oracle_verified_chunks"]}
    if "gather_ms_per_step_rank0" in r["config"]:
        out["gather_ms"] = r["config"]["gather_ms_per_step_rank0"]
    cb = r.get("cpu_baseline")
    if cb:
        out["cpu_GBps"] = cb["value"]; out["cpu_cores"] = cb["cores"]; out["cpu_socket_extrapolated"] = cb["linear_extrapolation_one_socket"]
        if cb["linear_extrapolation_one_socket"]:
            out["x_socket_extrapolated"] = round(r["value"] / cb["linear_extrapolation_one_socket"], 1)
    if name in LIGHT_WORKLOADS:   # (the line must stay well inside what the driver keeps of it: the headline's variants carry their rates only)
        out = {k: out[k] for k in ("value", "frac_step", "chunks", "encode_GBps", "decode_GBps", "top_kernel", "verified")}
    return {f"{name}_{k}": v for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chunks", type=int, default=None, help="chunks per GPU per step (default: 16 GiB of numbers per GPU)")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--gather", action="store_true", help="file assembly: compact on device, gather-v the chunk bytes to rank 0 over RCCL, scatter them back for the decode")
    ap.add_argument("--gather-carrier", default="cabi", choices=["cabi", "torch"], help="cabi: pco_gfx_gather_chunks / pco_gfx_scatter_chunks (the library calls RCCL itself); torch: pcodec_amd.sharding over torch.distributed")
    ap.add_argument("--verify-chunks", type=int, default=1024, help="chunks (drawn at random) whose bytes rank 0 compares with the oracle during warm-up")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--total-gib", type=float, default=0.0, help="STRONG scaling: this many GiB of numbers in total, split over the ranks in contiguous chunk blocks "
                    "(BASELINE configs[4]: --workload c5 --total-gib 64 --gather at 1/2/4/8 GPUs); default: 16 GiB per GPU, weak scaling")
    ap.add_argument("--no-others", action="store_true", help="only the headline workload (default: BASELINE's other configs are timed after it and attached as config.<name>_*)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)

    B = Bench(args)
    head = B.run(args.workload, args.chunks, args.steps, args.warmup, args.gather, args.verify_chunks, not args.no_cpu_baseline, carrier=args.gather_carrier)
    others = []
    if args.workload == "c2" and args.chunks is None and not args.no_others and not args.gather and not args.total_gib:
        for name, wl, chunks, steps, gather in OTHER_WORKLOADS:
            try:
                light = name in LIGHT_WORKLOADS
                # (three untimed calls: a spec's workspace is sized from what its first calls took -- an Auto call settles on its third, PCO_GFX_TRACE=1 -- and a
                #  timed step that still grows buffers measures hipMalloc, not the codec)
                r = B.run(wl, chunks, steps, 3, gather, min(args.verify_chunks, 64 if light else 256), not args.no_cpu_baseline and not gather and not light, cpu_seconds=6.0, carrier=args.gather_carrier)
            except Exception as e:   # an extra workload must not cost the headline its line
                if B.world > 1:
                    raise           # (one rank leaving a collective workload would hang the others)
                r = None; others.append((name, {"error": f"{type(e).__name__}: {e}"[:300]}))
            if r is not None:
                others.append((name, r))
    if B.rank == 0:
        roof = dict(head["roofline"]); direction = roof.pop("direction"); per_kernel = roof.pop("per_kernel_ms_per_step")
        for k, v in per_kernel.items():
            if v >= 0.02:
                roof[f"ms.{k}"] = v
        line = {"metric": "encode+decode GB/s (uncompressed) per chunk, u64/f64 2^18-elem", "value": head["value"], "unit": "GB/s",
                "n_gpus": B.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
                "higher_is_better": True, "scaling": "strong" if args.total_gib else "weak", "vs_baseline": None, "dtype": head["dtype"], "data": "synthetic",
                "config": dict(head["config"]), "roofline": roof}
        full = dict(line); full["config"] = dict(head["config"]); full["roofline"] = dict(head["roofline"])
        if others:
            line["config"]["other_workloads"] = ",".join(n for n, _ in others)
            full["config"]["other_workloads"] = []
            for name, r in others:
                if "error" in r:
                    line["config"][f"{name}_error"] = r["error"]; full["config"]["other_workloads"].append({"workload": name, **r}); continue
                line["config"].update(flat_workload(name, r))
                full["config"]["other_workloads"].append({"workload": name, "description": r["config"]["workload"], "dtype": r["dtype"], "value": r["value"], "unit": "GB/s",
                                                          "config": r["config"], "ms_per_step": r["ms_per_step"], "roofline": r["roofline"], "cpu_baseline": r.get("cpu_baseline")})
        if "cpu_baseline" in head:
            cb = head["cpu_baseline"]; full["cpu_baseline"] = cb
            line["cpu_baseline"] = {k: v for k, v in cb.items() if k not in ("single_thread", "cpu")}
            line["cpu_baseline"].update({"single_thread_enc_GBps": cb["single_thread"]["enc_gbs"], "single_thread_dec_GBps": cb["single_thread"]["dec_gbs"],
                                         "single_thread_both_GBps": cb["single_thread"]["both_gbs"], "cpu_model": cb["cpu"].get("model"),
                                         "cgroup_cpu_quota": cb["cpu"].get("cgroup_cpu_quota"), "cores_per_socket": cb["cpu"].get("cores_per_socket")})
            if cb.get("linear_extrapolation_one_socket"):
                line["cpu_baseline"]["gpu_over_socket_extrapolated"] = round(head["value"] / B.world / cb["linear_extrapolation_one_socket"], 1)
        line["config"] = ordered(line["config"], CONFIG_FIRST); line["roofline"] = ordered(line["roofline"], ROOFLINE_FIRST)
        full_s = json.dumps(full)
        print("BENCH_FULL " + full_s, file=sys.stderr, flush=True)
        try:
            with open(os.path.join(os.getcwd(), "bench_full.json"), "w") as f:
                f.write(full_s + "\n")
        except OSError:
            pass
        # the contract line must be the LAST thing on stdout: RCCL prints its version banner through C stdio, which (redirected to a file)
        # is only flushed at exit -- push it out first
        sys.stdout.flush()
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)
    if B.comm is not None:
        B.comm.close()
    if B.use_dist:
        B.dist.destroy_process_group()


if __name__ == "__main__":
    main()
