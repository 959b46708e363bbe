#!/bin/bash
# A/B of library builds on the bench: scripts/ab_libs.sh <chunks> <reps> lib1.so lib2.so ...  (per-kernel ms for the decode kernels)
CH=$1; REPS=$2; shift; shift
for r in $(seq $REPS); do for lib in "$@"; do
  PCO_GFX_LIB=$PWD/$lib timeout 300 python bench.py --steps 4 --warmup 1 --chunks $CH --no-cpu-baseline ${BENCH_ARGS} 2>&1 | grep metric | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); k=d['roofline']['per_kernel_avg_ms']; print('$lib'.ljust(28), d['value'], 'enc', d['config']['encode_GBps'], 'dec', d['config']['decode_GBps'], {a: round(b, 3) for a, b in k.items() if b > 0.5})"
done; done
