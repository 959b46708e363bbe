"""Turn a rocprofv3 rocpd database (--kernel-trace --stats) into a small per-kernel summary.
usage: python scripts/rocprof_summary.py <results.db> <out.md> [title]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc"))
regs = {}
for name, vg, sg, lds, gx, wx in cur.execute("select name, vgpr_count, sgpr_count, lds_size, grid_x, workgroup_x from kernels group by name"):
    regs[name] = (vg, sg, lds, gx, wx)
title = sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 --kernel-trace --stats"
with open(sys.argv[2], "w") as f:
    f.write(f"# {title}\n\n| kernel | calls | total (us) | average (us) | % | VGPR | SGPR | LDS (B) | grid_x x wg_x |\n|---|---|---|---|---|---|---|---|---|\n")
    for name, calls, tot, avg, pct in rows:
        short = name.split("(")[0].replace("void ", "")
        if "pcogfx" not in short and pct < 0.5:
            continue
        if len(short) > 70: short = short[:67] + "..."
        vg, sg, lds, gx, wx = regs.get(name, ("", "", "", "", ""))
        f.write(f"| `{short}` | {calls} | {tot:.1f} | {avg:.1f} | {pct:.2f} | {vg} | {sg} | {lds} | {gx} x {wx} |\n")
print(open(sys.argv[2]).read())
