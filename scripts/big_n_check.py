import sys, time; sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import numpy as np, oracle_lib as O, gpu_util as U
from pcodec_amd import _lib as G
rng=np.random.default_rng(3)
for n in (5_000_001, (1<<24) - 5):
    nums=(np.cumsum(rng.integers(-50,60,n))+(1<<40)).astype(np.int64)
    for kw in (dict(mode=1,delta=2,delta_order=1,max_page_n=1<<24), dict(mode=1,delta=1,max_page_n=1<<24), dict(mode=1,delta=2,delta_order=1)):
        t=time.time(); want=O.simple_compress(nums,O.make_config(**kw)); t1=time.time()
        got=U.gpu_simple_compress(nums,G.make_config(**kw)); t2=time.time()
        ok=got==want
        back=U.gpu_simple_decompress(got,nums.dtype,n)
        print(n,kw,"enc identical",ok,"dec ok",U.bits_equal(back,nums),"oracle %.1fs gpu %.2fs"%(t1-t,t2-t1),flush=True)
