#!/bin/bash
# lookback occupancy A/B on the Auto workloads (run on the GPU box)
for lib in "" ab/libpco_gfx_lbA.so ab/libpco_gfx_lbB.so; do
  for w in "c2auto 8192" "c3auto 8192" "c4 1024"; do set -- $w
    PCO_GFX_LIB=$lib timeout 300 python bench.py --workload $1 --chunks $2 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['per_kernel_ms_per_step']; print('${lib:-base}', '$1', d['value'], d['config']['encode_GBps'], {x:k[x] for x in k if 'lookback' in x})"
  done
done
