#!/bin/bash
# Bench lines of every workload on the GPU box: scripts/bench_all.sh <tag> [extra bench args]  -> gpurun_out/<tag>/bench_<workload>.json
TAG=${1:-r02}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { name=$1; shift; timeout 900 python bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "== $name rc=$?"; tail -c 400 $OUT/bench_$name.err; python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_$name.json").read().strip().splitlines()[-1])
    r = d["roofline"]; c = d["config"]
    print("$name", "value", d["value"], "enc", c["encode_GBps"], "dec", c["decode_GBps"], "C/chunk", c["compressed_bytes_per_chunk"], "dir", {k: (v["kernel_ms"], v["frac"]) for k, v in r["direction"].items()}, "cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("single_thread"), c.get("gather_ms_per_step_rank0"))
    print("   ", {k: v for k, v in r["per_kernel_ms_per_step"].items() if v > 0.3})
except Exception as e:
    print("$name", "no line:", e)
PY
}
run c2 --steps 10 --warmup 2 "$@"
run c5 --workload c5 --steps 5 --warmup 1 "$@"
run c5gather --workload c5 --gather --steps 5 --warmup 1 --no-cpu-baseline "$@"
run c3 --workload c3 --steps 5 --warmup 1 "$@"
run c1 --workload c1 --steps 5 --warmup 1 --chunks 8192 "$@"
run c4 --workload c4 --steps 3 --warmup 1 --chunks 1024 "$@"
run c2auto --workload c2auto --steps 5 --warmup 2 --chunks 8192 --no-cpu-baseline "$@"
run c3auto --workload c3auto --steps 5 --warmup 2 --chunks 8192 --no-cpu-baseline "$@"
run c5auto --workload c5auto --steps 5 --warmup 2 --no-cpu-baseline "$@"
