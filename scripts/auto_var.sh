#!/bin/bash
mkdir -p gpurun_out/r02c
for i in 1 2 3 4 5; do
PCO_GFX_TRACE=1 timeout 300 python bench.py --workload c2auto --chunks 8192 --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/r02c/var$i.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run$i', d['value'], d['ms_per_step'])"
grep "trace" gpurun_out/r02c/var$i.err | tail -13 | awk '{printf "%s %s %s %s | ", $3,$4,$5,$(NF-1)} END {print ""}'
done
