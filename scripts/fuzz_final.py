import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import fuzz_util
for seed in (404, 505, 606):
    bad, skipped, fails = fuzz_util.run(1500, seed, max_level=12)
    print("single", seed, "bad", len(bad), "skipped", sum(skipped.values()), bad[:3])
for seed in (11, 12):
    r = fuzz_util.run_batched(12, seed, max_level=12)
    print("batched", seed, r if not isinstance(r, tuple) else (len(r[0]), r[1] if len(r) > 1 else None))
