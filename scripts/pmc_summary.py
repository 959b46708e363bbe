"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database (pcogfx kernels only)."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set); dur = collections.defaultdict(float); grid = {}
for name, ctr, val, disp, d, gx, wx in cur.execute("select kernel_name, counter_name, value, dispatch_id, duration, grid_size_x, workgroup_size_x from counters_collection"):
    if "pcogfx" not in name: continue
    short = name.split("(")[0].replace("void ", "").replace("pcogfx::", "")
    acc[short][ctr] += val
    if disp not in cnt[short]: dur[short] += d
    cnt[short].add(disp); grid[short] = (gx, wx)
ctrs = sorted({c for k in acc for c in acc[k]})
print("kernel".ljust(60), "calls", "avg_us".rjust(9), " ".join(c.replace("SQ_", "").rjust(14) for c in ctrs))
for k in sorted(acc, key=lambda k: -dur[k]):
    n = len(cnt[k])
    print(k[:60].ljust(60), str(n).rjust(5), f"{dur[k] / n / 1e3:9.1f}", " ".join(f"{acc[k][c] / n:14.4g}" for c in ctrs), grid[k])
