"""Model of the tile-parallel Lomuto pass of encode_hist_literal.hip (lit_partition): 64 steps of sort_utils.rs:109-126 at a time in closed form
(prefix count of the smaller elements, C(q) = v[q] / v[q-1] / C(L(q)), in-tile pointer jumping, deferred write of the block front), checked
against the literal loop on random arrays, tile widths 4 / 8 / 64.  usage: python scripts/lomuto_tile_model.py"""
import numpy as np
def literal(v, pivot):
    v = v.copy(); left = 0
    for pos in range(len(v)):
        value = v[pos]; lt = value < pivot
        v[pos] = v[left]; v[left] = value; left += int(lt)
    return v, left
def tiled(v, pivot, W=64):
    A = v.copy(); n = len(A); left0 = 0
    prev_val = None; prev_lt = False
    P = 0
    while P < n:
        m = min(W, n - P)
        val = A[P:P+m].copy()
        lt = val < pivot
        rank = np.concatenate([[0], np.cumsum(lt)[:-1]]).astype(int)
        k = int(lt.sum())
        Lq = left0 + rank
        q = P + np.arange(m)
        C = np.zeros(m, dtype=A.dtype); ptr = np.arange(m)
        for l in range(m):
            if Lq[l] == q[l]:
                C[l] = val[l]
            else:
                pl = lt[l-1] if l > 0 else prev_lt
                pv = val[l-1] if l > 0 else prev_val
                if not pl: C[l] = pv
                else:
                    if Lq[l] >= P: ptr[l] = Lq[l] - P
                    else: C[l] = A[Lq[l]]
        for _ in range(6):
            C = C[ptr]; ptr = ptr[ptr]
        left1 = left0 + k
        # writes
        for l in range(m):
            if lt[l]: A[left0 + rank[l]] = val[l]
        for l in range(m):
            if q[l] >= left1: A[q[l]] = C[l]
        prev_val = val[m-1]; prev_lt = bool(lt[m-1])
        left0 = left1; P += m
    if n > 0 and not prev_lt and left0 < n:
        A[left0] = prev_val
    return A, left0
rng = np.random.default_rng(0)
for it in range(20000):
    n = int(rng.integers(0, 400))
    hi = int(rng.choice([2, 5, 100, 10**6]))
    v = rng.integers(0, hi, n)
    pivot = int(rng.integers(0, hi + 1))
    a, la = literal(v, pivot); b, lb = tiled(v, pivot, W=int(rng.choice([4, 8, 64])))
    assert la == lb and (a == b).all(), (it, n, v, pivot, a, b)
print("ok")
