#!/bin/bash
# enc_lookback_pipe_kernel: c4 encode against the number of page slots (pages in flight)
for S in 256 384 512 640 768 1024; do
  for W in "c4 1024" "c4 4096"; do
    set -- $W
    PCO_GFX_LB_SLOTS=$S python bench.py --workload $1 --chunks $2 --steps 2 --warmup 1 --no-cpu-baseline --no-others --verify-chunks 32 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); k=d['roofline']['per_kernel_ms_per_step']
print('slots=$S $1 $2: enc', d['config']['encode_GBps'], {x:k[x] for x in k if 'lookback' in x})"
  done
done
