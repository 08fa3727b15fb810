"""Structural constants of the MSK144 air interface, checked independently of the reference source."""
import numpy as np

from msk144cudecoder_amd import protocol as P
from msk144cudecoder_amd import synth


def _gf2_rank(M):
    M = M.copy() % 2
    r = 0
    rows, cols = M.shape
    for c in range(cols):
        piv = next((i for i in range(r, rows) if M[i, c]), None)
        if piv is None:
            continue
        M[[r, piv]] = M[[piv, r]]
        for i in range(rows):
            if i != r and M[i, c]:
                M[i] ^= M[r]
        r += 1
        if r == rows:
            break
    return r


def test_tanner_graph_structure():
    H = synth.ldpc_parity_matrix()
    assert H.shape == (38, 128)
    assert set(H.sum(0)) == {3}                                   # every bit sits in 3 checks
    deg = H.sum(1)
    assert sorted(np.nonzero(deg == 11)[0].tolist()) == [2, 4, 5, 26]   # the reference's is_full_row set
    assert set(deg) == {10, 11}
    assert H.sum() == 384
    assert _gf2_rank(H) == 38
    assert _gf2_rank(H[:, 90:]) == 38                             # parity block invertible -> systematic encoder exists
    for row in P.CHECK_BITS:
        bits = [b for b in row if b >= 0]
        assert bits == sorted(bits) and len(set(bits)) == len(bits)  # slots ascend with the bit index


def test_sync_word_crc_patterns():
    assert P.SYNC8 == [0, 1, 1, 1, 0, 0, 1, 0]
    assert P.CRC13_POLY == 0x15D7
    assert P.PATTERN_MASK[:6] == [[1] * (i + 1) + [0] * (5 - i) for i in range(6)]   # nested prefixes
    assert P.PATTERN_MASK[6] == [1, 0, 0, 1, 0, 0] and P.PATTERN_MASK[7] == [1, 0, 0, 1, 1, 0]
    assert P.PATTERN_NUM_AVG == [sum(m) for m in P.PATTERN_MASK]


def test_frequency_grid():
    assert P.grid(500, 1) == (501, -250.0)      # BASELINE deep config
    assert P.grid(100, 2) == (51, -50.0)        # README "optimal scan"
    assert P.grid(200, 2) == (101, -100.0)      # built-in defaults
    assert P.grid(0, 2)[0] == 1
    assert P.grid(7, 2) == (3, -2.0)            # half = int(1.75) = 1
    assert [P.clamp_scan_depth(d) for d in (-3, 0, 1, 8, 9)] == [1, 1, 1, 8, 8]


def test_encoder_and_modulator():
    rng = np.random.default_rng(0)
    H = synth.ldpc_parity_matrix()
    for _ in range(10):
        m = synth.random_message(rng)
        cw = synth.encode_message(m)
        assert np.array_equal(cw[:77], m)
        assert not (H.astype(int) @ cw % 2).any()
        x = synth.modulate_frame(synth.frame_bits(cw))
        assert x.shape == (864,)
        assert np.allclose(np.abs(x), 1.0, atol=1e-12)             # constant envelope
