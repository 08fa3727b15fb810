"""An independent float64 numpy reading of the reference's scan / softbits / BP semantics,
used only to cross-check the C++ oracle for coding slips (index arithmetic, tap alignment, edge order).
It shares no code with oracle/ and none with the HIP kernels; constants come from the protocol header."""
import numpy as np

from msk144cudecoder_amd import protocol as P

N = 5184
PP = np.sin(np.arange(12) * np.pi / 12.0)
S8 = 2 * np.array(P.SYNC8) - 1


def cb42():
    cbq = np.concatenate([PP[6:] * S8[0], PP * S8[2], PP * S8[4], PP * S8[6]])
    cbi = np.concatenate([PP * S8[1], PP * S8[3], PP * S8[5], PP[:6] * S8[7]])
    return cbi + 1j * cbq


def mix(cdat, freq):
    """exp(i*phi) * cdat with the reference's float32 phase quantisation (scan_kernel.cuh:54)."""
    n = np.arange(N, dtype=np.float32)
    f0 = np.float32(-freq)
    phi = ((n * np.float32(2.0 * np.float32(np.pi))) * f0) / np.float32(12000.0)
    return np.exp(1j * phi.astype(np.float64)) * cdat.astype(np.complex128)


def scan_xb(cdat2, mask):
    """xb for all 5376 scanned positions of one pattern (scan_kernel.cuh:87-124)."""
    cb = cb42()
    pos = np.arange(5376)
    k = np.arange(42)
    s = np.zeros(5376, dtype=np.complex128)
    y = np.zeros((5376, 42), dtype=np.complex128)
    for m in range(6):
        if mask[m]:
            y += cdat2[(pos[:, None] + k[None, :] + 864 * m) % N]
            y += cdat2[(pos[:, None] + k[None, :] + 864 * m + 336) % N]
    s = (np.conj(y) * cb[None, :]).sum(axis=1)
    return np.abs(s)


def slots_from_xb(xb):
    """top-8-of-21-slice-maxima rule with the reference's tie breaks (scan_kernel.cuh:140-353)."""
    slot_xb = np.zeros(8)
    slot_pos = np.zeros(8, dtype=np.int64)
    for s in range(21):
        seg = xb[256 * s:256 * (s + 1)]
        j = int(np.argmax(seg))               # first maximum = lowest position
        best, best_pos = seg[j], 256 * s + j
        w = int(np.argmin(slot_xb))           # first minimum = lowest slot
        if best > slot_xb[w]:
            slot_xb[w], slot_pos[w] = best, best_pos
    return slot_pos, slot_xb


def softbits(cdat2, mask, pos):
    """144 raw softbits, 128 LLRs and nbadsync of one candidate (softbits_kernel.cuh:56-247)."""
    cb = cb42()
    n = np.arange(864)
    c3 = np.zeros(864, dtype=np.complex128)
    for m in range(6):
        if mask[m]:
            c3 += cdat2[(pos + n + 864 * m) % N]
    s = (c3[:42] * np.conj(cb)).sum() + (c3[336:378] * np.conj(cb)).sum()
    c3 = c3 * np.exp(-1j * np.angle(s))
    soft = np.zeros(144)
    for piq in range(72):
        i_idx = (12 * piq + np.arange(12)) % 864
        q_idx = (858 + 12 * piq + np.arange(12)) % 864
        soft[2 * piq + 1] = (c3.real[i_idx] * PP).sum()
        soft[2 * piq] = (c3.imag[q_idx] * PP).sum()
    sav = soft.mean()
    s2av = (soft ** 2).mean()
    ssig = np.sqrt(s2av - sav * sav)
    scale = 2.0 / (ssig * 0.6 * 0.6)
    llr = scale * np.concatenate([soft[8:56], soft[64:144]])
    hard = np.where(soft < 0, -1, 1)
    nbad = int(((8 - (hard[0:8] * S8).sum()) // 2) + ((8 - (hard[56:64] * S8).sum()) // 2))
    return soft, llr, nbad


def platanh(x):
    z = abs(x)
    sgn = -1.0 if x < 0 else 1.0
    if z <= 0.664:
        return x / 0.83
    if z <= 0.9217:
        return sgn * (z - 0.4064) / 0.322
    if z <= 0.9951:
        return sgn * (z - 0.8378) / 0.0524
    if z <= 0.9998:
        return sgn * (z - 0.9914) / 0.0012
    return sgn * 7.0


def crc13_ok(cw):
    bits = list(cw[:77]) + [0] * 19
    rem = 0
    for b in bits:
        rem = (rem << 1) | int(b)
        if rem & 0x2000:
            rem ^= 0x2000 | P.CRC13_POLY
    rx = 0
    for b in cw[77:90]:
        rx = (rx << 1) | int(b)
    return (rem & 0x1FFF) == rx


def bp_decode(llr):
    """(accepted, message, iteration, hard errors) - ldpc_kernel.cuh:100-249, check-major graph walk."""
    checks = [[n for n in row if n >= 0] for row in P.CHECK_BITS]
    edges = [(c, n) for c, row in enumerate(checks) for n in row]
    tov = {e: 0.0 for e in edges}
    by_bit = {n: [e for e in edges if e[1] == n] for n in range(128)}
    for it in range(10):
        zn = np.array([llr[n] + sum(tov[e] for e in by_bit[n]) for n in range(128)])
        cw = (zn > 0).astype(int)
        synd = sum(sum(cw[n] for n in row) % 2 for row in checks)
        nhard = int(sum((cw[n] == 1) != (llr[n] > 0) if cw[n] == 1 else not (llr[n] <= 0) for n in range(128)))
        if synd == 0 and crc13_ok(cw) and nhard < 18:
            return True, cw[:77], it, nhard
        toc = {e: zn[e[1]] - tov[e] for e in edges}
        th = {e: np.tanh(-0.5 * toc[e]) for e in edges}
        for c, row in enumerate(checks):
            for n in row:
                prod = 1.0
                for n2 in row:
                    if n2 != n:
                        prod *= th[(c, n2)]
                tov[(c, n)] = 2.0 * platanh(-prod)
    return False, None, None, None
