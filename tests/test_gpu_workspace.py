"""The library's device scratch (pco_gfx_workspace_bytes): full-width latent scratch goes to the chunks that write full-width latents and to
no others, and whichever way a chunk gets there -- never speculating, taken out by the 64-position sample, failing in the 16-bit split --
its bytes are the oracle's."""
import numpy as np
import pytest

import gpu_util as U
import oracle_lib as O
from pcodec_amd import _lib as G

pytestmark = pytest.mark.gpu
N18 = U.N18


@pytest.fixture(scope="module")
def L():
    return G.lib()


def _check_against_oracle(arrays, chunks, back, ocfg):
    for a, c, b in zip(arrays, chunks, back):
        assert U.bits_equal(a, b)
        assert c == U.chunk_of_file(O.simple_compress(a, ocfg), len(c))


def test_full_width_scratch_goes_to_the_chunks_that_need_it(L):
    pcfg, ocfg = U.cfg_pair("c2")
    narrow = [U.synth("c2", seed=100 + s) for s in range(30)]
    in_bytes = sum(a.nbytes for a in narrow)
    slot = (N18 + 256) * 8   # one chunk's full-width latents (one variable)

    L.pco_gfx_release_workspace()
    assert L.pco_gfx_workspace_bytes() == 0
    chunks, back = U.gpu_batched(narrow, pcfg)
    _check_against_oracle(narrow[:4], chunks[:4], back[:4], ocfg)
    ws_narrow = L.pco_gfx_workspace_bytes()
    # 16-bit latents (4 B reserved per number), symbols and tANS fields (3 B), decode symbols (3 B) + per-chunk state, an eighth of slack on
    # every buffer, one full-width slot that is always there: well below two bytes per input byte (it was 2.3 with a slot per chunk)
    assert ws_narrow < 1.7 * in_bytes, ws_narrow / in_bytes

    # the same call with three chunks that need full-width latents, each for another reason
    r = np.random.default_rng(7)
    wide = r.integers(0, 1 << 62, N18, dtype=np.uint64)                 # the sample sees it: taken out before the split
    spike = U.synth("c2", seed=501); spike[100_003] += np.uint64(1 << 50)   # the sample does not: a tile of the 16-bit split fails, the chunk is redone
    spike2 = U.synth("c2", seed=502); spike2[7] ^= np.uint64(1 << 63)
    mixed = narrow[:27] + [wide, spike, spike2]
    order = r.permutation(len(mixed))
    mixed = [mixed[i] for i in order]
    L.pco_gfx_release_workspace()
    chunks, back = U.gpu_batched(mixed, pcfg)
    _check_against_oracle(mixed, chunks, back, ocfg)
    ws_mixed = L.pco_gfx_workspace_bytes()
    # `wide` also needs the two sort buffers of the wide-range histogram, which are per call (2 x 8 B per number of every chunk)
    sort_bytes = len(mixed) * 2 * slot
    extra = ws_mixed - ws_narrow
    assert 2 * slot <= extra - sort_bytes * 9 // 8 <= 4 * slot * 9 // 8 + (1 << 20), (extra, sort_bytes, slot)

    # 32-bit chunks: full-width scratch is 4 bytes per number
    L.pco_gfx_release_workspace()
    pc1, oc1 = U.cfg_pair("c1")
    u32 = [U.synth("c1", seed=s) for s in range(8)]
    chunks, back = U.gpu_batched(u32, pc1)
    _check_against_oracle(u32, chunks, back, oc1)
    ws32 = L.pco_gfx_workspace_bytes()
    assert ws32 < 7.0 * sum(a.nbytes for a in u32), ws32 / sum(a.nbytes for a in u32)   # (8.5 with 8-byte elements)
    L.pco_gfx_release_workspace()


def test_every_chunk_fails_the_split_and_none_the_sample(L):
    """All chunks of a call are redone at full width (an outlier the sample cannot see): the first hand-out is empty, the second takes a
    slot per chunk.  Lookback and float-mult calls next to it: slots from the start, two variables per slot."""
    pcfg, ocfg = U.cfg_pair("c2")
    arrs = []
    for s in range(6):
        a = U.synth("c2", seed=900 + s); a[50_001 + 4099 * s] += np.uint64(1 << 45); arrs.append(a)
    L.pco_gfx_release_workspace()
    chunks, back = U.gpu_batched(arrs, pcfg)
    _check_against_oracle(arrs, chunks, back, ocfg)
    for kind in ("c4", "c3", "c3d"):
        p, o = U.cfg_pair(kind)
        arrs = [U.synth(kind, n=70_000 + 1000 * s, seed=s) for s in range(5)]
        if kind != "c4":   # a float that is no multiple of the base: the secondary variable leaves 16 bits in one chunk
            arrs[2] = arrs[2].copy(); arrs[2][33_333] = 1e300
        chunks, back = U.gpu_batched(arrs, p)
        _check_against_oracle(arrs, chunks, back, o)
    L.pco_gfx_release_workspace()
