"""An input ORDER on which the reference's histogram takes its heapsort branch (histograms.rs:248-258), built the way McIlroy's
"A Killer Adversary for Quicksort" builds one: the literal algorithm (oracle/pco_oracle_encode.hpp restates it; this script restates its
control flow once more, on element identities) is run against an adversary that leaves every number undecided ("gas", larger than anything
decided) until a comparison needs it, and then decides it to be the next smallest value.  Every pivot choose_pivot (sort_utils.rs:5-56)
can find is therefore one of the smallest numbers of its range, every partition is lopsided (sort_utils.rs:109-126: bad), and after
1 + floor(log2(n + 1)) of them on one recursion path the branch runs.  The numbers still undecided then were only ever compared with
pivots: they can take ANY values above the decided ones -- here a few hundred distinct values, each repeated many times, so that runs of
equal numbers straddle the bin ends and apply_sorted's tie rule (histograms.rs:164-206) shows against the quickselect path's.

usage: python scripts/make_hist_fallback_fixture.py           (writes tests/golden/hist_fallback.npz: u32 arrays for n = 2^18 and n = 5000)
The construction is deterministic; tests/test_oracle_golden.py checks that the ORACLE reports the branch on the committed arrays."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


class Fallback(Exception):
    pass


def adversarial_order(n, n_bins_log=8):
    """-> (values decided by the adversary: dict position -> value, positions still gas, the range [start, start + len) that was heapsorted)"""
    v = list(range(n))            # v[k] = identity (original position) of the element now at index k
    val = {}                      # identity -> decided value
    nxt = [1]

    def freeze(i):
        val[i] = nxt[0]; nxt[0] += 1

    def lt(i, j):                 # value(i) < value(j), deciding as late and as small as possible
        a, b = val.get(i), val.get(j)
        if a is None and b is None:
            freeze(i); return True
        if a is None: return False
        if b is None: return True
        return a < b

    def choose_pivot(lo, ln):     # sort_utils.rs:5-56 on v[lo : lo + ln]; returns the INDEX (relative) of the chosen element
        a, b, c = ln // 4, ln // 2, (ln * 3) // 4
        if ln >= 8:
            def sort2(x, y): return (y, x) if lt(v[lo + y], v[lo + x]) else (x, y)
            def sort3(x, y, z):
                x, y = sort2(x, y); y, z = sort2(y, z); x, y = sort2(x, y); return x, y, z
            if ln >= 50:
                def adj(x): return sort3(x - 1, x, x + 1)[1]
                a, b, c = adj(a), adj(b), adj(c)
            a, b, c = sort3(a, b, c)
        return b

    def break_patterns(lo, ln):   # sort_utils.rs:61-105
        if ln >= 8:
            seed = ln; mask64 = (1 << 64) - 1
            modulus = 1
            while modulus < ln: modulus <<= 1
            pos = ln // 4 * 2
            for i in range(3):
                seed ^= (seed << 13) & mask64; seed ^= seed >> 7; seed ^= (seed << 17) & mask64
                other = seed & (modulus - 1)
                if other >= ln: other -= ln
                p, q = lo + pos - 1 + i, lo + other
                v[p], v[q] = v[q], v[p]

    n_bins = 1 << n_bins_log
    state = {"n_applied": 0}
    bin_idx = lambda c: (c << n_bins_log) // n
    c_count = lambda b: ((b + 1) * n + n_bins - 1) >> n_bins_log

    def recurse(lo, ln, lb, ub, limit):   # histograms.rs:208-280 (the bounds are (tight, x); values of gas elements never reach them)
        if ln == 0: return
        target_c = c_count(bin_idx(state["n_applied"]))
        if state["n_applied"] + ln <= target_c or lb[1] == ub[1] or ln == 1:
            state["n_applied"] += ln; return
        pi = choose_pivot(lo, ln)
        pid = v[lo + pi]
        if pid not in val: freeze(pid)     # (cannot happen: the median of three is never the undecided one; kept for safety)
        tentative = val[pid]
        if tentative > lb[1]: pivot, lhs_ub, rhs_lb = tentative, (False, tentative - 1), (True, tentative)
        else: pivot, lhs_ub, rhs_lb = tentative + 1, (True, tentative), (False, tentative + 1)
        left = 0
        for pos in range(lo, lo + ln):     # sort_utils.rs:109-126
            e = v[pos]; x = val.get(e)
            is_lt = x is not None and x < pivot
            v[pos] = v[lo + left]; v[lo + left] = e
            if is_lt: left += 1
        if 1 + min(left, ln - left) < ln // 8:
            limit -= 1
            if limit == 0: raise Fallback((lo, ln))
            break_patterns(lo, left); break_patterns(lo + left, ln - left)
        recurse(lo, left, lb, lhs_ub, limit)
        recurse(lo + left, ln - left, rhs_lb, ub, limit)

    limit = 1 + int(np.floor(np.log2(n + 1)))
    sys.setrecursionlimit(10000)
    try:
        recurse(0, n, (False, 0), (False, (1 << 32) - 1), limit)
    except Fallback as f:
        return val, f.args[0]
    raise RuntimeError("the adversary did not reach the heapsort branch")


def build(n, seed, rule_differs):
    """The adversary's order with the undecided numbers drawn from a few hundred values of very unequal frequency (long runs of equal
    numbers beside single ones: that is where apply_sorted's tie rule and the quickselect path's part).  Draws until `rule_differs`
    says the literal algorithm and the multiset rule disagree on the result."""
    val, (lo, ln) = adversarial_order(n)
    base = max(val.values()) + 16
    rng = np.random.default_rng(seed)
    for _ in range(1000):
        k = int(rng.integers(2, 400))
        w = rng.pareto(0.7, k) + 0.01; w /= w.sum()
        vals = np.sort(rng.choice(1 << 20, k, replace=False)).astype(np.uint32)
        x = (base + vals[rng.choice(k, n, p=w)]).astype(np.uint32)
        for i, ww in val.items():
            x[i] = ww
        if rule_differs(x):
            return x, (lo, ln), len(val)
    raise RuntimeError("no tie pattern found on which the two rules differ")


if __name__ == "__main__":
    import oracle_lib as O
    out = {}
    differs = lambda x: O.histogram(x.copy(), 8, rule=0)[0] != O.histogram(x.copy(), 8, rule=1)[0]
    for n, seed in ((1 << 18, 3), (5000, 4)):
        x, rng, decided = build(n, seed, differs)
        bins0, fb0 = O.histogram(x.copy(), 8, rule=0)                  # the literal algorithm, on the order as given
        bins1, fb1 = O.histogram(np.sort(x), 8, rule=1)               # the quickselect path's result as a function of the sorted multiset (what the GPU computes)
        same = bins0 == bins1
        print(f"n = {n}: {decided} numbers decided by the adversary, heapsorted range {rng}, oracle reports the branch: {fb0}; literal == multiset rule: {same} ({len(bins0)} vs {len(bins1)} bins)")
        assert fb0 and not same, "the oracle did not take the heapsort branch on this order, or the two rules agree on it"
        out[f"n{n}"] = x
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "hist_fallback.npz"), **out)
    print("wrote tests/golden/hist_fallback.npz")
