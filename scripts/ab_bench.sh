#!/bin/bash
# bench at several chunk counts, printing per-kernel averages
for ch in ${CHUNKS:-2048 4096}; do
  echo "== chunks=$ch"
  timeout 300 python bench.py --steps 4 --warmup 1 --chunks $ch --no-cpu-baseline 2>&1 | grep metric | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], 'GB/s enc', d['config']['encode_GBps'], 'dec', d['config']['decode_GBps'], d['roofline']['per_kernel_avg_ms'])"
done
