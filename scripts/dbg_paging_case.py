"""Replay of one tests/test_gpu_paging.py exact-paging case, page by page (which page list, which page, first differing byte)."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests")); sys.path.insert(0, os.path.join(HERE, ".."))
import oracle_lib as O
import test_gpu_paging as T
from pcodec_amd import _lib as G
import ctypes as C

case = int(sys.argv[1])
L = G.lib()
for f in ("pco_chunk_compressor_n_pages", "pco_chunk_compressor_page_n", "pco_chunk_compressor_meta_size", "pco_chunk_compressor_page_size", "pco_page_decompressor_consumed"):
    getattr(L, f).restype = C.c_size_t
kind, kw = T.EXACT_CASES[case]
rng = np.random.default_rng(1000 + case)
for sizes in T.PAGE_LISTS:
    n = sum(sizes)
    nums = T.data_for(kind, n, rng)
    kw8 = dict(kw, enable_8_bit=True)
    want_meta, want_pages, want_ns = O.wrapped_compress(nums, O.make_config(**kw8), exact_pages=sizes)
    meta, pages, page_ns = T.gpu_wrapped_exact(L, nums, G.make_config(**kw8), sizes)
    print(kind, kw, sizes, "meta", meta == want_meta, len(meta))
    for i, (a, b) in enumerate(zip(pages, want_pages)):
        if a != b:
            d = next((j for j in range(min(len(a), len(b))) if a[j] != b[j]), -1)
            nd = sum(1 for j in range(min(len(a), len(b))) if a[j] != b[j])
            print("  page", i, "n", sizes[i], "len", len(a), len(b), "first diff", d, "n diff", nd, a[max(0,d-4):d+12].hex(), b[max(0,d-4):d+12].hex())
