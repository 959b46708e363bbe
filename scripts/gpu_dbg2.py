"""Dump walker symbols for one chunk and compare with the symbols implied by the bins (classic, no delta)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle_lib as O
import gpu_util as U
rng = np.random.default_rng(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
nums = np.where(rng.random(n) < 0.7, rng.integers(0, 16, n), rng.integers(1000, 100000, n)).astype(np.uint32)
enc = O.simple_compress(nums, O.make_config(delta=1))
info, bins = O.inspect_first_chunk(enc)
print("asl", list(info.ans_size_log), "nbins", list(info.n_bins))
b = bins[1]
print("bins(weight,lower,ob):", b[:12].tolist())
lowers = b[:, 1].astype(np.int64)
order = np.argsort(lowers)
exp_sym = order[np.searchsorted(lowers[order], nums.astype(np.int64), side="right") - 1]
os.environ["PCO_GFX_DEBUG_DUMP"] = "/tmp/dump.bin"
got = U.gpu_simple_decompress(enc, nums.dtype, n)
print("decode ok:", U.bits_equal(got, nums))
raw = open("/tmp/dump.bin", "rb").read()
ss, os_ = np.frombuffer(raw[:16], np.uint64)
syms = np.frombuffer(raw[16:16 + 3 * int(ss)], np.uint8)[int(ss):2 * int(ss)]
# undo the block layout
out = np.zeros(n, np.int64)
for i in range(n):
    bt = i // 256; r = i % 256; g, j = r // 4, r % 4
    out[i] = syms[bt * 256 + 16 * (g // 4) + 4 * j + (g % 4)]
bad = np.nonzero(out != exp_sym)[0]
print("first bad", bad[:10], "n bad", len(bad))
print("got ", out[:40].tolist()); print("exp ", exp_sym[:40].tolist())
op = np.frombuffer(raw[16 + 3 * int(ss):], np.uint64)[int(os_):int(os_) + 4]
print("offpos", op.tolist())
