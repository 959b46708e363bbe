"""The hand-over between the publishing tANS walker and the expander kernel that runs under it (decode_trail.hip) when the two are NOT
co-resident: an expander wave that sees no walker for ~55 ms leaves, and every chunk the walker marked for the expanders that no expander
wave finished (DecPlan::fused without the done mark in the block's progress line) is expanded by dec_expand_kernel afterwards and counted
(pco_gfx_trail_givebacks).  Round 4's form could lose such a chunk silently (status OK, stale bytes) when the walker started after the
expander's initial wait had expired, and whenever the expander kernel refused a chunk the walker had marked.

The switches are read once per process, so every scenario runs in a child: PCO_GFX_TRAIL_DEBUG=d launches the expanders BEFORE the walkers
(all of them time out in their initial wait, then the walkers mark and publish to nobody), =n launches no expanders at all."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu


def small_mixed_blocks():
    """Walker blocks of eight 16-bit chunks each: classic candidates for the expanders next to int-mult chunks of a handful of numbers
    (fewer than 24 bytes of chunk: the per-block glance at the preambles does not see them as two-variable chunks, the walker marks them
    for the second expander kernel, the first refuses them)."""
    import torch
    import gpu_util as U
    import oracle_lib as O
    from pcodec_amd import _lib as G
    rng = np.random.default_rng(5)
    arrays, cfgs = [], []
    for rep in range(16):
        for slot in range(8):
            if slot in (2, 5):
                n = int(rng.integers(2, 6)); a = (rng.integers(0, 3, n) * 7).astype(np.int16); kw = dict(mode=4, mode_u64=7, delta=1)
            else:
                n = int(rng.choice([300, 700, 1000, 257])); a = (np.arange(n) * 3 + rng.integers(0, 40, n)).astype(np.uint16); kw = dict(mode=1, delta=2, delta_order=1)
            arrays.append(a); cfgs.append(kw)
    chunks, back = [], []
    files = [O.simple_compress(a, O.make_config(**kw)) for a, kw in zip(arrays, cfgs)]
    import test_gpu_parity as T
    blobs = [U.chunk_of_file(f, len(f) - T.O_header_len(f) - 1) for f in files]
    k = len(blobs)
    import ctypes as C
    d_src = [torch.from_numpy(np.frombuffer(b + b"\0" * 16, np.uint8).copy()).cuda() for b in blobs]
    outs = [torch.full((a.nbytes + 16,), 0xAB, dtype=torch.uint8, device="cuda") for a in arrays]
    dt = (G.DecodeTask * k)(*[G.DecodeTask(d_src[i].data_ptr(), len(blobs[i]), outs[i].data_ptr(), arrays[i].size, G.DTYPE_BYTE[arrays[i].dtype.name], 0) for i in range(k)])
    dr = (G.TaskResult * k)()
    G.check(G.lib().pco_gfx_decompress_chunks(k, dt, dr, None, None))
    for i, a in enumerate(arrays):
        assert dr[i].status == 0 and dr[i].n_out == a.size, (i, dr[i].status, dr[i].n_out)
        got = outs[i][: a.nbytes].cpu().numpy().view(a.dtype)
        assert U.bits_equal(got, a), (i, cfgs[i], a.size, got[:8], a[:8])
        assert bool((outs[i][a.nbytes:] == 0xAB).all()), i
    return k


def child(body, **env):
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from pcodec_amd import _lib as G\nimport test_gpu_trail_handover as H, test_gpu_parity as T\n" % (os.path.join(HERE, ".."), HERE)) + body
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return r.stdout


@pytest.mark.parametrize("switch", ["d", "n"])
def test_chunks_nobody_expanded_under_the_walk_are_expanded_after_it(switch):
    """The whole mixed batch of test_decode_expanders_under_the_walk (2560 chunks of every kind, ragged, damaged ones among them, then 9000
    chunks through the persistent grid) and the small mixed blocks, with the expanders too early / absent: bit-exact, and the library
    reports how many chunks were given back."""
    out = child("T.test_decode_expanders_under_the_walk(None)\nprint('after the mixed batch', G.lib().pco_gfx_trail_givebacks(), 'marked', G.lib().pco_gfx_trail_marked())\nk = H.small_mixed_blocks()\nprint('givebacks', G.lib().pco_gfx_trail_givebacks(), k, 'marked', G.lib().pco_gfx_trail_marked())\n",
                PCO_GFX_TRAIL_DEBUG=switch, PCO_GFX_DEC_TRAIL="2")
    gb = int(out.split("givebacks")[1].split()[0]); marked = int(out.split("marked")[-1].split()[0])
    # every chunk the walker marked for the expanders came back (all but the handful of damaged ones, which report their error instead); none was lost
    assert marked >= 9800 and marked - 8 <= gb <= marked, out


def test_small_mixed_blocks_with_the_expanders_running():
    """The same small blocks with both kernels side by side: the chunks the walker marks for an expander kernel that refuses them (tiny
    two-variable chunks in a block the glance took for classic) are the only ones given back."""
    out = child("k = H.small_mixed_blocks()\nprint('givebacks', G.lib().pco_gfx_trail_givebacks(), k)\n", PCO_GFX_DEC_TRAIL="2")
    gb, k = (int(x) for x in out.split("givebacks")[1].split()[:2])
    assert gb <= k // 4, out


def test_decode_under_contention_stays_bit_exact():
    """A second process keeps the CUs busy with LDS-heavy kernels (rocBLAS GEMMs through torch) while this one decodes the mixed batch
    again and again: whatever the dispatcher does to the co-residency of walker and expanders, every number comes back, and the stall
    is visible in pco_gfx_trail_givebacks() (reported, not asserted: how the two processes interleave is not deterministic)."""
    hog = subprocess.Popen([sys.executable, "-c",
                            "import torch, time\na = torch.randn(4096, 4096, device='cuda')\nt = time.time()\n"
                            "while time.time() - t < 25:\n    for _ in range(20): a = (a @ a) * 1e-4\n    torch.cuda.synchronize()\n"],
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        out = child("import time\nt = time.time(); n = 0\nwhile time.time() - t < 12 or n < 2:\n    T.test_decode_expanders_under_the_walk(None); n += 1\n"
                    "print('givebacks', G.lib().pco_gfx_trail_givebacks(), n)\n")
        print(out.strip().splitlines()[-1])
    finally:
        hog.kill(); hog.wait()
