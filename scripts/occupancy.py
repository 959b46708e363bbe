"""Resident blocks per CU as the HIP runtime computes them (needs a -DPCO_OCCUPANCY_PROBE build: PCO_GFX_LIB=ab/libpco_gfx_occ.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pcodec_amd import _lib as G
L = G.lib()
for which, name in enumerate(["enc_hist_select_kernel", "enc_hist_wide_kernel<16384>", "enc_walk_kernel<8>", "enc_pack_kernel"]):
    b = C.c_int(0); rc = L.pco_gfx_debug_occupancy(which, C.byref(b)); print(name, "rc", rc, "blocks/CU", b.value)
bad = C.c_uint(99); rc = L.pco_gfx_debug_xor_lane_check(C.byref(bad)); print("xor_lane check rc", rc, "mismatches", bad.value)
