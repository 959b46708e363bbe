#!/usr/bin/env python3
"""bench.py -- whole-job encode+decode throughput of the pcodec chunk hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic chunks that are already resident
in HBM: pco_gfx_compress_chunks over every chunk, then pco_gfx_decompress_chunks over the chunks
just produced.  Workload = BASELINE.json configs[1]: u64, classic mode, delta order 1, 2^18-element
chunks of a noisy linear ramp.  value = uncompressed GB/s over encode+decode, 2*bytes/(t_enc+t_dec),
aggregated over all ranks (chunks are independent: each rank owns its own chunks, weak scaling, no
collective on the data path; `--gather` adds the optional RCCL gather of the compressed pages).

For N>1 the driver launches one rank per GPU through torch.distributed.run (RCCL).  PyTorch is only
plumbing here (device buffers, streams, process group); the codec is libpco_gfx.so.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N18 = 1 << 18
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

ENC_TASK = np.dtype([("src", "<u8"), ("n", "<u8"), ("dst", "<u8"), ("dst_cap", "<u8"), ("dtype", "<u4"), ("reserved", "<u4")])
DEC_TASK = np.dtype([("src", "<u8"), ("src_len", "<u8"), ("dst", "<u8"), ("dst_cap", "<u8"), ("dtype", "<u4"), ("flags", "<u4")])
RESULT = np.dtype([("n_out", "<u8"), ("consumed", "<u8"), ("status", "<u4"), ("aux", "<u4")])

WORKLOADS = {
    # name: (torch dtype name, pco dtype byte, config kwargs, description)
    "c2": ("int64", 2, dict(mode=1, delta=2, delta_order=1), "u64 classic delta-1 noisy ramp, 2^18-element chunks (BASELINE configs[1])"),
    "c3": ("float64", 6, dict(mode=2, mode_f64=0.01, delta=1), "f64 float-mult(0.01) decimals, 2^18-element chunks (BASELINE configs[2])"),
    "c1": ("int32", 1, dict(mode=1, delta=1), "u32 classic no-delta uniform random, 2^18-element chunks (BASELINE configs[0], incompressible)"),
    "c4": ("int64", 4, dict(mode=1, delta=3), "i64 seasonal (period 365) lookback delta, 2^18-element chunks (BASELINE configs[3])"),
    "c2auto": ("int64", 2, dict(), "u64 noisy ramp, default ChunkConfig (Auto mode + Auto delta), 2^18-element chunks"),
}
ELEM_BYTES = {"c1": 4, "c2": 8, "c3": 8, "c4": 8, "c2auto": 8}


def make_chunks(torch, kind, n_chunks, rank, device):
    g = torch.Generator(device=device)
    g.manual_seed(1234 + 7919 * rank)
    if kind in ("c2", "c2auto"):
        i = torch.arange(N18, device=device, dtype=torch.int64)
        base = (1 << 40) + 1000 * i
        noise = torch.randint(0, 512, (n_chunks, N18), generator=g, device=device, dtype=torch.int64)
        start = torch.randint(0, 1 << 20, (n_chunks, 1), generator=g, device=device, dtype=torch.int64)
        return (base.unsqueeze(0) + noise + start).contiguous()  # non-negative: same bits as u64
    if kind == "c3":
        cents = torch.randint(1000, 10000, (n_chunks, N18), generator=g, device=device, dtype=torch.int64)
        return (cents.to(torch.float64) / 100.0).contiguous()
    if kind == "c1":
        return torch.randint(-(1 << 31), 1 << 31, (n_chunks, N18), generator=g, device=device, dtype=torch.int64).to(torch.int32).contiguous()
    if kind == "c4":
        base = torch.randint(-(1 << 40), 1 << 40, (365,), generator=g, device=device, dtype=torch.int64)
        idx = torch.arange(N18, device=device) % 365
        return (base[idx].unsqueeze(0) + torch.randint(-3, 4, (n_chunks, N18), generator=g, device=device, dtype=torch.int64)).contiguous()
    raise KeyError(kind)


def cpu_baseline(kind, cfg_kw, seconds=12.0):
    """The oracle (a port of the reference's algorithm, NOT the Rust binary) on the host cores."""
    import oracle_lib as O
    import gpu_util as U

    nums = U.synth(kind)
    ocfg = O.make_config(**cfg_kw)
    enc = O.simple_compress(nums, ocfg)
    cores = os.cpu_count() or 1
    # single thread, a few chunks
    t0 = time.perf_counter(); k1 = 0
    te = td = 0.0
    while time.perf_counter() - t0 < seconds / 3 or k1 < 2:
        a = time.perf_counter(); O.simple_compress(nums, ocfg); b = time.perf_counter(); O.simple_decompress(enc, nums.dtype, cap=nums.size + 8); c = time.perf_counter()
        te += b - a; td += c - b; k1 += 1
    one = dict(enc_gbs=k1 * nums.nbytes / te / 1e9, dec_gbs=k1 * nums.nbytes / td / 1e9, both_gbs=2 * k1 * nums.nbytes / (te + td) / 1e9)
    # all cores, one chunk stream per thread (ctypes releases the GIL)
    counts = [0] * cores
    stop = time.perf_counter() + 2 * seconds / 3

    def work(i):
        local = nums.copy()
        while time.perf_counter() < stop:
            e = O.simple_compress(local, ocfg); O.simple_decompress(e, local.dtype, cap=local.size + 8); counts[i] += 1

    ts = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    t1 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    el = time.perf_counter() - t1
    total = sum(counts)
    return {"value": round(2 * total * nums.nbytes / el / 1e9, 3), "unit": "GB/s", "cores": cores, "kind": "port",
            "sample": f"{total} chunks of 2^18 {nums.dtype.name} ({kind}) encode+decode over {cores} threads in {el:.1f}s "
                      f"(C++ restatement of the reference, g++ -O3 -mavx2; the Rust reference cannot be built here)",
            "single_thread": {k: round(v, 3) for k, v in one.items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chunks", type=int, default=8192, help="chunks per GPU per step (8192 x 2 MiB = 16 GiB of numbers per GPU)")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--gather", action="store_true", help="also gather the compressed pages to rank 0 over RCCL each step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pcodec_amd import _lib as G

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: pcodec_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    L = G.lib()

    tdt, dtb, cfg_kw, desc = WORKLOADS[args.workload]
    gcfg = G.make_config(**cfg_kw)
    nch = args.chunks
    data = make_chunks(torch, args.workload, nch, rank, device)
    chunk_bytes = N18 * ELEM_BYTES[args.workload]
    cap = (L.pco_gfx_guarantee_chunk_size(N18, dtb) + 64 + 15) // 16 * 16
    comp = torch.zeros(nch * cap, dtype=torch.uint8, device=device)
    out = torch.empty_like(data)

    enc_tasks = np.zeros(nch, ENC_TASK)
    enc_tasks["src"] = data.data_ptr() + np.arange(nch, dtype=np.uint64) * chunk_bytes
    enc_tasks["n"] = N18; enc_tasks["dtype"] = dtb; enc_tasks["dst_cap"] = cap
    enc_tasks["dst"] = comp.data_ptr() + np.arange(nch, dtype=np.uint64) * cap
    dec_tasks = np.zeros(nch, DEC_TASK)
    dec_tasks["src"] = enc_tasks["dst"]; dec_tasks["dst"] = out.data_ptr() + np.arange(nch, dtype=np.uint64) * chunk_bytes
    dec_tasks["dst_cap"] = N18; dec_tasks["dtype"] = dtb
    enc_res = np.zeros(nch, RESULT); dec_res = np.zeros(nch, RESULT)

    def encode():
        G.check(L.pco_gfx_compress_chunks(nch, enc_tasks.ctypes.data, C.byref(gcfg), enc_res.ctypes.data, None, None))

    def decode():
        dec_tasks["src_len"] = enc_res["n_out"]
        G.check(L.pco_gfx_decompress_chunks(nch, dec_tasks.ctypes.data, dec_res.ctypes.data, None, None))

    def gather_pages():
        # optional: rank 0 collects every rank's compressed chunks, in chunk order, over RCCL (pcodec_amd/sharding.py)
        if world == 1:
            return
        from pcodec_amd import sharding as S
        sizes = torch.from_numpy(enc_res["n_out"].astype(np.int64)).to(device)
        n_out = enc_res["n_out"]
        payload = torch.cat([comp[i * cap: i * cap + int(n_out[i])] for i in range(nch)])   # chunks back to back
        S.gather_pages(payload, sizes, dst=0)

    def step():
        encode()
        if args.gather:
            gather_pages()
        decode()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # warm-up, with the reference bench's bitwise round-trip assertion (pco_cli/src/bench/codecs/mod.rs:176-189)
    for w in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    assert torch.equal(out, data), "decode(encode(x)) != x"
    if rank == 0:
        import oracle_lib as O
        first = data[0].cpu().numpy().view({"c1": np.uint32, "c2": np.uint64, "c3": np.float64, "c4": np.int64, "c2auto": np.uint64}[args.workload])
        want = O.simple_compress(first, O.make_config(**cfg_kw))
        got = bytes(comp[: int(enc_res["n_out"][0])].cpu().numpy())
        hdr = len(want) - 1 - len(got)
        assert got == want[hdr:-1], "GPU chunk bytes differ from the oracle"

    # timed region: exactly K steps, bracketed by barrier + synchronize
    sync_all()
    L.pco_gfx_profile_begin()
    t_enc = t_dec = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        a = time.perf_counter(); encode()
        if args.gather:
            gather_pages()
        b = time.perf_counter(); decode(); c = time.perf_counter()
        t_enc += b - a; t_dec += c - b
    sync_all()
    elapsed = time.perf_counter() - t0
    names = C.create_string_buffer(1 << 16); ms = (C.c_float * 4096)()
    nk = L.pco_gfx_profile_end(names, len(names), ms, 4096)
    el = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    ms_per_step = elapsed * 1e3 / args.steps
    total_bytes = world * nch * chunk_bytes
    value = 2 * total_bytes / (ms_per_step * 1e-3) / 1e9

    if rank == 0:
        raw = names.raw; kn = []; pos = 0
        for _ in range(nk):
            e = raw.index(b"\0", pos); kn.append(raw[pos:e].decode()); pos = e + 1
        per = {}
        for nm, t in zip(kn, ms[:nk]):
            per.setdefault(nm, []).append(float(t))
        kavg = {k: sum(v) / len(v) for k, v in per.items()}
        comp_bytes = int(enc_res["n_out"].sum())
        # algorithmic bytes per launch: encode = n*sizeof(T) read + C written; decode = C read + n*sizeof(T) written
        alg = nch * chunk_bytes + comp_bytes
        dom = max(kavg, key=kavg.get)
        traffic = None
        try:  # HBM bytes of the dominant kernel from the committed PMC passes (profiles/), scaled to this run's chunk count
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if tj.get("workload") == args.workload and dom in tj["kernels"]:
                kk = tj["kernels"][dom]
                traffic = int((kk["fetch_bytes_per_launch"] + kk["write_bytes_per_launch"]) * nch / tj["chunks"])
        except (OSError, ValueError, KeyError):
            pass
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(alg / (kavg[dom] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(alg / (kavg[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": alg, "avg_launch_ms": round(kavg[dom], 4),
                "per_kernel_avg_ms": {k: round(v, 4) for k, v in sorted(kavg.items())}}
        line = {
            "metric": "encode+decode GB/s (uncompressed) per chunk, u64/f64 2^18-elem", "value": round(value, 2), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {"c1": "u32", "c2": "u64", "c3": "f64", "c4": "i64", "c2auto": "u64"}[args.workload],
            "data": "synthetic",
            "config": {"workload": desc, "chunks_per_gpu": nch, "chunk_n": N18, "compression_level": 8,
                       "mode_spec": {"c3": "TryFloatMult(0.01)", "c2auto": "Auto"}.get(args.workload, "Classic"),
                       "delta_spec": {"c1": "NoOp", "c2": "TryConsecutive(1)", "c3": "NoOp", "c4": "TryLookback", "c2auto": "Auto"}[args.workload],
                       "parallelism": f"chunk-sharded x{world}" + (" + RCCL gather of pages" if args.gather else ""),
                       "compressed_bytes_per_chunk": comp_bytes // nch,
                       "encode_GBps": round(world * nch * chunk_bytes * args.steps / t_enc / 1e9, 2),
                       "decode_GBps": round(world * nch * chunk_bytes * args.steps / t_dec / 1e9, 2)},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.workload, cfg_kw)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
