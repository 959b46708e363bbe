"""One line per bench.py JSON line on stdin: value, encode / decode GB/s and the kernels above a threshold (ms per step)."""
import json
import sys

thr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
tag = sys.argv[2] if len(sys.argv) > 2 else ""
for l in sys.stdin:
    if not l.startswith("{"):
        continue
    d = json.loads(l); r = d["roofline"]
    ks = {k[3:]: v for k, v in r.items() if k.startswith("ms.") and v >= thr}
    print(tag, "value", d["value"], "enc", d["config"]["encode_GBps"], "dec", d["config"]["decode_GBps"],
          "kernel_ms enc", r.get("kernel_ms_encode"), "dec", r.get("kernel_ms_decode"), ks, flush=True)
