#!/bin/bash
# A/B of the decode paths on the GPU box: PCO_GFX_DEC_TRAIL=1 (default: expanders under the walk for calls of >= 1024 chunks), 0 (walk kernel, then expand kernel)
for F in 1 0; do
  for W in "c2 8192" "c3 8192" "c5 12288" "c1 16384"; do
    set -- $W
    PCO_GFX_DEC_TRAIL=$F python bench.py --workload $1 --chunks $2 --steps 5 --warmup 1 --no-cpu-baseline --no-others --verify-chunks 64 2>/dev/null | python scripts/bench_brief.py 0.05 "trail=$F $1 $2:"
  done
done
