"""csrc/mix.h: the kernels' own sincos (Cody-Waite reduction by pi + minimax polynomials), emulated operation by operation in
float32 with exact-product FMAs and compared with double precision over the phase range the mixers use (|phi| < 6000 rad).
CPU only: the constants are parsed from the header, so the test follows the code."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "msk144cudecoder_amd", "csrc", "mix.h")).read()
BODY = SRC[SRC.index("void sincos_reduced("):SRC.index("// x / 12000 correctly rounded")]
F32 = np.float32


def fma(a, b, c):
    """float32 fused multiply-add: the product of two float32 is exact in float64, one rounding of the sum."""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(F32)


def constants():
    lits = [F32(x) for x in re.findall(r"(-?\d+\.\d+(?:e-?\d+)?)f", BODY)]
    k_round, inv_pi, pi_hi, pi_lo = lits[0], lits[1], lits[2], lits[3]
    s4, s3, s2, s1 = lits[4:8]
    c5, c4, c3, c2, c1, one = lits[8:14]
    assert k_round == F32(12582912.0) and one == F32(1.0) and c1 == F32(-0.5)
    assert abs(float(inv_pi) - 1 / np.pi) < 1e-7 and abs(float(pi_hi) + float(pi_lo) - np.pi) < 1e-14
    return k_round, inv_pi, pi_hi, pi_lo, (s1, s2, s3, s4), (c1, c2, c3, c4, c5)


def sincos_model(phi):
    k_round, inv_pi, pi_hi, pi_lo, s, c = constants()
    phi = phi.astype(F32)
    kb = fma(phi, inv_pi, k_round)
    k = (kb - k_round).astype(F32)
    r = fma(-k, pi_hi, phi)
    r = fma(-k, pi_lo, r)
    z = (r * r).astype(F32)
    sp = fma(s[3], z, s[2])
    sp = fma(sp, z, s[1])
    sp = fma(sp, z, s[0])
    s_r = fma((r * z).astype(F32), sp, r)
    cp = fma(c[4], z, c[3])
    cp = fma(cp, z, c[2])
    cp = fma(cp, z, c[1])
    cp = fma(cp, z, c[0])
    c_r = fma(cp, z, F32(1.0))
    flip = (kb.view(np.uint32) << np.uint32(31))
    return (s_r.view(np.uint32) ^ flip).view(F32), (c_r.view(np.uint32) ^ flip).view(F32), r


def test_sincos_matches_double_precision_to_two_ulp_of_one():
    rng = np.random.default_rng(7)
    phi = np.concatenate([rng.uniform(-6000.0, 6000.0, 2_000_000), np.linspace(-10.0, 10.0, 200_001),
                          np.arange(-1900, 1901) * np.pi, (np.arange(-1900, 1901) + 0.5) * np.pi]).astype(F32)
    sn, cs, r = sincos_model(phi)
    assert np.abs(r).max() <= np.pi / 2 * 1.001                      # the reduction lands inside the fitted interval
    d = phi.astype(np.float64)                                        # the float phase IS the argument (as for the reference's sincosf)
    assert np.abs(sn - np.sin(d)).max() < 2.0e-7
    assert np.abs(cs - np.cos(d)).max() < 2.0e-7
    assert np.abs(sn.astype(np.float64) ** 2 + cs.astype(np.float64) ** 2 - 1.0).max() < 5e-7
