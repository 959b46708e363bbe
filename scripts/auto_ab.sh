#!/bin/bash
# Auto-spec traces (run on the GPU box): scripts/auto_ab.sh [chunks]
CH=${1:-8192}
mkdir -p gpurun_out/r02c
run() { PCO_GFX_TRACE=1 timeout 300 python bench.py --workload $2 --chunks $3 --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/r02c/$1.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['config']['encode_GBps'], d['config']['decode_GBps']); print({k:v for k,v in d['roofline']['per_kernel_ms_per_step'].items() if v>1.5})"; grep "mode\|resolve\|final\|trial enc" gpurun_out/r02c/$1.err | tail -9; }
run c2auto c2auto $CH
run c3auto c3auto $CH
run c5auto c5auto $CH
