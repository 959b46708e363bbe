"""The exactness argument of enc_walkseg_kernel (pcodec_amd/csrc/encode_walkseg.hip), checked on the CPU by brute force.

The kernel starts a segment of the reverse tANS walk (ans/encoding.rs:65-87) from the ARC of all states, steps the arc's two ends, and takes
the state for known once the ends are equal.  Here the same two-ended step (ws_step2, restated in Python) runs beside ALL T trajectories on
random tables -- any weights, any assignment of table positions to symbols (the reference's spread is one of them: what the argument needs is
only that a symbol's row of next states ascends with its index, as encoding.rs builds it) -- and at every step
  * every state any start could have reached lies on the arc from A to B (the arc is a superset), and
  * when A == B every start has reached exactly that state.
A table that never forgets (two or four equal bins under the reference's spread: some trajectories stay apart for ever) must never claim
to have met."""
import numpy as np
import pytest


def make_table(rng, weights, asl, spread="random"):
    T = 1 << asl
    assert sum(weights) == T
    if spread == "reference":   # ans/spec.rs:37-58: position (stride * step) mod T, stride = 3 T / 5 made odd
        stride = (T * 3 // 5) | 1
        pos = [(stride * k) % T for k in range(T)]
    else:
        pos = list(rng.permutation(T))
    owner = np.empty(T, np.int64); k = 0
    for s, w in enumerate(weights):
        for _ in range(w): owner[pos[k]] = s; k += 1
    rows = [T + np.nonzero(owner == s)[0] for s in range(len(weights))]   # next_states[s][j], ascending in j (encoding.rs pushes table_size + state_idx in table order)
    return rows


def step(x, w, row):
    bits = 0
    while (x >> bits) >= 2 * w: bits += 1
    return int(row[(x >> bits) - w]), (x >> bits) - w


def step2(a, b, w, row, T):
    """ws_step2: an arc past the 2T-1 -> T cut whose ends reach the same row entry covers the whole row and becomes the arc of all states first."""
    _, ka = step(a, w, row); _, kb = step(b, w, row)
    if b < a and ka == kb: a, b = T, 2 * T - 1
    return step(a, w, row)[0], step(b, w, row)[0]


def on_arc(x, a, b):
    return a <= x <= b if a <= b else (x >= a or x <= b)


def random_weights(rng, T, n_bins):
    cuts = np.sort(rng.choice(np.arange(1, T), n_bins - 1, replace=False)) if n_bins > 1 else np.array([], int)
    return [int(v) for v in np.diff(np.concatenate([[0], cuts, [T]]))]


@pytest.mark.parametrize("asl", [2, 4, 6, 8])
def test_the_arc_contains_every_trajectory_and_meets_only_when_all_have(asl):
    rng = np.random.default_rng(100 + asl)
    T = 1 << asl
    met_some = 0
    for trial in range(60):
        n_bins = int(rng.integers(1, min(T, 40) + 1))
        weights = random_weights(rng, T, n_bins)
        rows = make_table(rng, weights, asl, "reference" if trial % 3 == 0 else "random")
        p = np.array(weights) / T
        syms = rng.choice(n_bins, 200, p=p) if trial % 2 == 0 else rng.integers(0, n_bins, 200)   # by weight, or uniformly (rare symbols often)
        states = np.arange(T, 2 * T)
        a, b = T, 2 * T - 1
        for s in syms:
            w, row = weights[s], rows[s]
            states = np.array([step(int(x), w, row)[0] for x in np.unique(states)])
            a, b = step2(a, b, w, row, T)
            assert all(on_arc(int(x), a, b) for x in states), (asl, trial, weights)
            if a == b:
                assert len(np.unique(states)) == 1 and int(states[0]) == a, (asl, trial, weights)
                met_some += 1
                break
    assert met_some > 20   # (most random tables forget within 200 symbols)


def test_a_table_that_never_forgets_never_claims_to():
    """Two or four equal bins under the reference's spread: the states a row can leave keep landing on different entries of the next row, and
    some trajectories stay apart for ever (brute force: 2 to 16 of them after any number of steps) -- the two ends of the arc must then never
    become equal, however long the segment.  (The kernel walks such a segment again from the true state: the old chain, a segment at a time.)"""
    rng = np.random.default_rng(7)
    for asl, n_bins in ((4, 4), (6, 2), (8, 4), (10, 4)):
        T = 1 << asl
        weights = [T // n_bins] * n_bins
        rows = make_table(rng, weights, asl, "reference")
        states = np.arange(T, 2 * T); a, b = T, 2 * T - 1
        for s in rng.integers(0, n_bins, 600):
            w, row = weights[s], rows[s]
            states = np.unique(np.array([step(int(x), w, row)[0] for x in states]))
            a, b = step2(a, b, w, row, T)
            assert len(states) >= 2, (asl, n_bins)
            assert a != b
            assert all(on_arc(int(x), a, b) for x in states)
