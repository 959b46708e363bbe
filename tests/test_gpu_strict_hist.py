"""PCO_GFX_CFG_STRICT_HISTOGRAM: the literal replay of the reference's quickselect histogram on the device (encode_hist_literal.hip;
histograms.rs:60-298, sort_utils.rs).  In strict mode the GPU's bytes are the reference's on EVERY input order -- the orders that send
the reference into its heapsort branch (histograms.rs:248-258) included, which the default histogram kernels do not reproduce
(tests/test_gpu_parity.py::test_the_heapsort_branch_of_the_reference_histogram says what they write there)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import gpu_util as U
import oracle_lib as O
from pcodec_amd import _lib as G

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "scripts"))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    lib = G.lib()
    if lib.pco_gfx_device_count() < 1:
        pytest.fail("no HIP device: pcodec_amd has no CPU fallback")
    return lib


def adversarial(n, seed, dt):
    """A fresh order of McIlroy's adversary (scripts/make_hist_fallback_fixture.py) against the reference's pivot rule: the ~100 numbers
    the quickselect ever compares are the smallest, everything else is drawn from a few hundred values of very unequal frequency."""
    import make_hist_fallback_fixture as M
    val, _ = M.adversarial_order(n)
    rng = np.random.default_rng(seed)
    base = max(val.values()) + 16
    span = (1 << 14) if np.dtype(dt).itemsize == 2 else (1 << 20)
    for _ in range(60):   # (draw until apply_sorted's tie rule and the quickselect path's give different histograms, if that comes soon)
        k = int(rng.integers(2, 400))
        w = rng.pareto(0.7, k) + 0.01; w /= w.sum()
        vals = np.sort(rng.choice(span, k, replace=False)).astype(np.int64)
        x = base + vals[rng.choice(k, n, p=w)]
        for i, ww in val.items():
            x[i] = ww
        u = x.astype(np.uint32)
        if O.histogram(u.copy(), 8, rule=0)[0] != O.histogram(u.copy(), 8, rule=1)[0]: break
    if np.dtype(dt) == np.int16: return (x - 32768).astype(np.int16)          # (order-preserving: the latent is x ^ 0x8000)
    if np.dtype(dt) == np.uint64: return (x + (1 << 40)).astype(np.uint64)
    return x.astype(dt)


def test_strict_mode_writes_the_reference_bytes_on_the_heapsort_branch(L):
    """Both committed adversarial orders and 27 fresh ones (u32 / u64 / i16 x n = 5000 / 70 000 / 2^18 x 3 tie patterns): the oracle reports
    the branch, the strict GPU bytes equal the LITERAL reference bytes, the device reports that it replayed the branch, and the default
    mode still writes the multiset rule's bytes (the documented divergence, unchanged)."""
    kw = dict(mode=1, delta=1)
    fx = np.load(os.path.join(HERE, "golden", "hist_fallback.npz"))
    cases = [(f"fixture {k}", fx[k]) for k in ("n5000", "n262144")]
    for dt in (np.uint32, np.uint64, np.int16):
        for n in (5000, 70000, 1 << 18):
            for seed in range(3):
                cases.append((f"{np.dtype(dt).name} n={n} seed={seed}", adversarial(n, 1000 * seed + n % 997, dt)))
    differing = 0
    for name, x in cases:
        literal = O.simple_compress(x, O.make_config(**kw))
        _, _, fb = O.chunk_plan(x, O.make_config(**kw))
        assert fb, f"{name}: the oracle did not take the heapsort branch"
        before = L.pco_gfx_strict_histogram_fallbacks()
        got = U.gpu_simple_compress(x, G.make_config(strict_histogram=True, **kw))
        assert got == literal, f"{name}: strict GPU bytes differ from the reference's"
        assert L.pco_gfx_strict_histogram_fallbacks() == before + 1, f"{name}: the device did not replay the heapsort branch"
        O.set_hist_rule(1)
        try:
            multiset = O.simple_compress(x, O.make_config(**kw))
        finally:
            O.set_hist_rule(0)
        assert U.gpu_simple_compress(x, G.make_config(**kw)) == multiset, name
        differing += literal != multiset
        assert U.bits_equal(U.gpu_simple_decompress(got, x.dtype, x.size), x)
    assert differing >= 15, f"only {differing} of {len(cases)} orders tell the two tie rules apart"


def test_strict_mode_equals_the_oracle_on_every_baseline_config(L):
    """Strict mode where the branch does not run: the replay's bins are the fast kernels' bins, so every BASELINE config at 2^18 stays
    byte-identical to the oracle (and the replay reports no fallback)."""
    before = L.pco_gfx_strict_histogram_fallbacks()
    for kind in ("c1", "c2", "c3", "c3d", "c4", "auto"):
        nums = U.synth(kind if kind != "auto" else "c2")
        gcfg, ocfg = U.cfg_pair(kind)
        gcfg.flags = G.CFG_STRICT_HISTOGRAM
        assert U.gpu_simple_compress(nums, gcfg) == O.simple_compress(nums, ocfg), kind
    assert L.pco_gfx_strict_histogram_fallbacks() == before


def test_strict_mode_randomised_sweep(L):
    """The randomised sweep of test_gpu_parity.py (dtype x size x distribution x mode x delta x level x paging) with the strict flag:
    every case is compared -- the skip for the reference's order-dependent branch does not exist here -- single and batched calls,
    levels up to 12, Auto mode / Auto delta (whose trial encodes histogram their samples the same way) included."""
    import fuzz_util
    bad, skipped, _ = fuzz_util.run(260, 4242, max_level=12, strict=True)
    assert not bad, bad[:10]
    assert all(k.startswith("oracle refused") for k in skipped), skipped
    bad, skipped, _ = fuzz_util.run(120, 4243, only_8bit=True, strict=True)
    assert not bad, bad[:10]
    assert all(k.startswith("oracle refused") for k in skipped), skipped
    bad, skipped = fuzz_util.run_batched(6, 78, max_level=12, strict=True)
    assert not bad, bad[:10]


def test_strict_mode_through_the_environment(L):
    """PCO_GFX_STRICT_HISTOGRAM=1 turns the flag on for callers of the reference's own three-function C ABI, whose PcoChunkConfig has no
    field for it: checked in a child process (the variable is read once per process)."""
    import subprocess
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import gpu_util as U, oracle_lib as O\nfrom pcodec_amd import _lib as G\n"
            "x = np.load(%r)['n5000']\n"
            "assert U.gpu_simple_compress(x, G.make_config(mode=1, delta=1)) == O.simple_compress(x, O.make_config(mode=1, delta=1))\n"
            "assert G.lib().pco_gfx_strict_histogram_fallbacks() == 1\nprint('ok')\n") % (os.path.join(HERE, ".."), HERE, os.path.join(HERE, "golden", "hist_fallback.npz"))
    env = dict(os.environ, PCO_GFX_STRICT_HISTOGRAM="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
