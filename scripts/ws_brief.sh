#!/bin/bash
# workspace figure and throughput of a few workloads: scripts/ws_brief.sh "c2 8192" "c3 8192" ...
for w in "$@"; do set -- $w; python bench.py --workload $1 --chunks $2 --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline --no-others --verify-chunks 64 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; r=d['roofline']
        print(c['workload'][:24], 'value', d['value'], 'enc', c['encode_GBps'], 'dec', c['decode_GBps'], 'workspace_x', c['workspace_bytes_per_input_byte'], 'kernel_ms', r['kernel_ms_encode'], r['kernel_ms_decode'])
"; done
