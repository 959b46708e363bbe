#!/bin/bash
# A/B of the lookback search kernels on the GPU box: PCO_GFX_LB_PIPE=1 (five-wave pipeline) vs 0 (one wave per page)
OUT=gpurun_out/${1:-lb_ab}; mkdir -p $OUT
for P in 1 0; do
  for W in "c4 1024" "c4 4096" "c2auto 8192" "c3auto 8192"; do
    set -- $W
    PCO_GFX_LB_PIPE=$P python bench.py --workload $1 --chunks $2 --steps 3 --warmup 1 --no-cpu-baseline --no-others --verify-chunks 64 > $OUT/${1}_${2}_pipe$P.json 2> $OUT/${1}_${2}_pipe$P.err
    python - <<PY
import json
try:
    d=json.load(open("$OUT/${1}_${2}_pipe$P.json"))
    k=d["roofline"]["per_kernel_ms_per_step"]
    print("pipe=$P $1 $2: value", d["value"], "enc", d["config"]["encode_GBps"], "dec", d["config"]["decode_GBps"], {x:k[x] for x in k if "lookback" in x})
except Exception as e:
    print("pipe=$P $1 $2 FAILED", e); print(open("$OUT/${1}_${2}_pipe$P.err").read()[-1500:])
PY
  done
done
