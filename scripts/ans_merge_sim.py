"""tANS state coalescence under the reference encoder (ans/spec.rs spread, ans/encoding.rs:65-87): how many symbols until two start states meet."""
import numpy as np
def tables(weights, asl):
    T=1<<asl; assert sum(weights)==T
    step=(T*3//5)|1
    ss=np.zeros(T,np.int64); pos=0
    for s,w in enumerate(weights):
        for _ in range(w): ss[pos]=s; pos=(pos+step)%T
    idx=[np.nonzero(ss==s)[0] for s in range(len(weights))]
    return idx
def enc_step(x, s, w, idx, T):
    # x in [T,2T)
    bits=0
    while (x>>bits) >= 2*w: bits+=1
    return T+idx[s][(x>>bits)-w]
def trial(weights, asl, rng, nsteps=4000):
    T=1<<asl; idx=tables(weights,asl)
    p=np.array(weights)/T
    res=[]
    for t in range(300):
        a=T; b=T+int(rng.integers(0,T))
        syms=rng.choice(len(weights), nsteps, p=p)
        k=0
        while a!=b and k<nsteps:
            s=syms[k]; w=weights[s]
            a=enc_step(a,s,w,idx,T); b=enc_step(b,s,w,idx,T); k+=1
        res.append(k)
    r=np.array(res); return np.median(r), np.percentile(r,90), np.percentile(r,99), r.max()


def full_merge(weights, asl, rng, nsteps=6000, trials=300):
    """Steps until the trajectories from the lowest and the highest state meet under random symbols drawn by weight: an upper bound of what
    enc_walkseg_kernel walks twice per segment (its arc of all states shrinks at least as fast)."""
    T = 1 << asl; idx = tables(weights, asl); p = np.array(weights) / T; res = []
    for _ in range(trials):
        a = T; b = 2 * T - 1
        syms = rng.choice(len(weights), nsteps, p=p); k = 0
        while a != b and k < nsteps:
            s = syms[k]; w = weights[s]; a = enc_step(a, s, w, idx, T); b = enc_step(b, s, w, idx, T); k += 1
        res.append(k)
    r = np.array(res); return float(np.median(r)), float(np.percentile(r, 90)), float(np.percentile(r, 99)), int(r.max())


if __name__ == "__main__":
    # the bench workloads' own tables (through the oracle): python scripts/ans_merge_sim.py
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    import oracle_lib as O, gpu_util as U
    rng = np.random.default_rng(3)
    for w in ["c2", "c3", "c4"]:
        nums = U.synth(w); gcfg, ocfg = U.cfg_pair(w)
        info, bins = O.inspect_first_chunk(O.simple_compress(nums, ocfg))
        for v in range(3):
            if info.var_present[v] and info.n_bins[v] > 1:
                wts = [int(x) for x in bins[v][:, 0]]
                print(w, "variable", v, "ans_size_log", int(info.ans_size_log[v]), "bins", len(wts), "steps until every state has met (median, p90, p99, max of 300):", full_merge(wts, int(info.ans_size_log[v]), rng))
    for name, wts, asl in (("256 equal bins", [4] * 256, 10), ("64 equal bins", [16] * 64, 10), ("16 equal bins", [64] * 16, 10), ("4 equal bins", [256] * 4, 10)):
        print(name, full_merge(wts, asl, rng, trials=100))
