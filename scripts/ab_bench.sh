#!/bin/bash
# A/B: waves-per-SIMD cap x chunks per launch
for lib in libpco_gfx.so libpco_gfx_w2.so; do
  for ch in 2048 4096 8192; do
    echo "== $lib chunks=$ch"
    PCO_GFX_LIB=$PWD/pcodec_amd/$lib timeout 300 python bench.py --steps 4 --warmup 1 --chunks $ch --no-cpu-baseline 2>&1 | grep metric | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], 'GB/s enc', d['config']['encode_GBps'], 'dec', d['config']['decode_GBps'], d['roofline']['per_kernel_avg_ms'])"
  done
done
