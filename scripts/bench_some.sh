#!/bin/bash
# scripts/bench_some.sh <tag> <workload:args>...   e.g.  scripts/bench_some.sh t1 "c1:--chunks 8192" "c5:"
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for spec in "$@"; do
  name=${spec%%:*}; args=${spec#*:}
  wl=${name%%_*}
  timeout 900 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline $args > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "== $name rc=$?"; tail -c 300 $OUT/bench_$name.err | grep -v amdgpu.ids
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_$name.json").read().strip().splitlines()[-1])
    r = d["roofline"]; c = d["config"]
    print("$name", "value", d["value"], "enc", c["encode_GBps"], "dec", c["decode_GBps"], "C/chunk", c["compressed_bytes_per_chunk"], "dir", {k: (v["kernel_ms"], v["frac"]) for k, v in r["direction"].items()})
    print("   ", {k: v for k, v in r["per_kernel_ms_per_step"].items() if v > 0.25})
except Exception as e:
    print("$name", "no line:", e)
PY
done
