"""MSK144 signal synthesiser: CRC-13, (128,90) LDPC encoder, MSK modulator, AWGN streams.

The reference ships no transmitter and its only sample recording is absent, so every input used by
the tests and by bench.py is generated here.  The signal model is the inverse of the reference's
demodulator:

* frame = s8 | cw[0:48] | s8 | cw[48:128]                        (softbits_kernel.cuh:204-211)
* odd bits ride on I with a 12-sample half-sine, even bits on Q shifted by 6 samples
                                                                 (softbits_kernel.cuh:158-177,
                                                                  msk_context.cuh:188-196)
* cw = 77 message bits | 13 CRC bits | 38 parity bits            (ldpc_kernel.cuh:45-63)

Nothing in this module is taken from the oracle; tests use it to drive both the oracle and the HIP
decoder with the same samples.
"""
from __future__ import annotations

import dataclasses
from typing import Iterable, Sequence

import numpy as np

from .protocol import (CHECK_BITS, CRC13_POLY, FRAME_SAMPLES, SAMPLE_RATE, SYNC8, WINDOW_SAMPLES)

__all__ = [
    "crc13_bits", "ldpc_parity_matrix", "ldpc_encode", "encode_message", "frame_bits", "modulate_frame",
    "random_message", "Ping", "synth_audio", "synth_iq", "stream_s1", "stream_s2", "stream_s3", "pack_bits_msb", "iq_low_snr_batch",
]

_PP = np.sin(np.arange(12) * np.pi / 12.0)


# --------------------------------------------------------------------------------------------
# CRC-13 (polynomial 0x15D7): 77 bits -> 12 bytes MSB first, bits 77..95 zero, bytewise CRC.
# Bit-serial long division here (independent of the table-driven form the decoder uses).
# --------------------------------------------------------------------------------------------
def crc13_bits(msg77: Sequence[int]) -> np.ndarray:
    bits = np.zeros(96, dtype=np.uint8)
    bits[:77] = np.asarray(msg77, dtype=np.uint8) & 1
    rem = 0
    for b in bits:
        rem = ((rem << 1) | int(b)) & 0x3FFF
        if rem & 0x2000:
            rem ^= (0x2000 | CRC13_POLY)
    # The decoder's table walk shifts 8 bits per byte with the register pre-loaded one byte late:
    # after 12 bytes the remainder equals the plain long division of the 96-bit block.
    return np.array([(rem >> (12 - i)) & 1 for i in range(13)], dtype=np.uint8)


def ldpc_parity_matrix() -> np.ndarray:
    """H as a (38,128) 0/1 matrix from the check-major Tanner graph."""
    H = np.zeros((38, 128), dtype=np.uint8)
    for c, row in enumerate(CHECK_BITS):
        for n in row:
            if n >= 0:
                H[c, n] = 1
    return H


def _gf2_inv(A: np.ndarray) -> np.ndarray:
    n = A.shape[0]
    M = np.concatenate([A.copy() % 2, np.eye(n, dtype=np.uint8)], axis=1)
    for col in range(n):
        piv = next((r for r in range(col, n) if M[r, col]), None)
        if piv is None:
            raise ValueError("parity block is singular")
        if piv != col:
            M[[col, piv]] = M[[piv, col]]
        for r in range(n):
            if r != col and M[r, col]:
                M[r] ^= M[col]
    return M[:, n:]


_GEN = None


def _generator() -> np.ndarray:
    """P (38x90) with parity = P @ m90 (mod 2): solves H[:,90:] p = H[:,:90] m."""
    global _GEN
    if _GEN is None:
        H = ldpc_parity_matrix()
        inv = _gf2_inv(H[:, 90:])
        _GEN = (inv.astype(np.int64) @ H[:, :90].astype(np.int64)) % 2
    return _GEN


def ldpc_encode(m90: Sequence[int]) -> np.ndarray:
    m = np.asarray(m90, dtype=np.int64) & 1
    p = (_generator() @ m) % 2
    return np.concatenate([m, p]).astype(np.uint8)


def encode_message(msg77: Sequence[int]) -> np.ndarray:
    """77 message bits -> 128-bit codeword."""
    msg = np.asarray(msg77, dtype=np.uint8) & 1
    return ldpc_encode(np.concatenate([msg, crc13_bits(msg)]))


def frame_bits(cw128: Sequence[int]) -> np.ndarray:
    cw = np.asarray(cw128, dtype=np.uint8)
    s8 = np.asarray(SYNC8, dtype=np.uint8)
    return np.concatenate([s8, cw[:48], s8, cw[48:]])


def modulate_frame(bits144: Sequence[int]) -> np.ndarray:
    """144 bits -> 864 complex baseband samples, constant envelope 1."""
    d = 2.0 * np.asarray(bits144, dtype=np.float64) - 1.0
    i_part = np.zeros(FRAME_SAMPLES)
    q_part = np.zeros(FRAME_SAMPLES)
    for j in range(72):
        i_part[12 * j:12 * j + 12] += d[2 * j + 1] * _PP
        idx = (12 * j - 6 + np.arange(12)) % FRAME_SAMPLES
        q_part[idx] += d[2 * j] * _PP
    return i_part + 1j * q_part


def random_message(rng: np.random.Generator, i3: int = 1) -> np.ndarray:
    """77 random bits with the 3-bit message type forced (i3=1: standard message)."""
    m = rng.integers(0, 2, size=77, dtype=np.uint8)
    m[74:77] = [(i3 >> 2) & 1, (i3 >> 1) & 1, i3 & 1]
    return m


def pack_bits_msb(bits: Sequence[int]) -> bytes:
    """bits -> bytes, MSB first, zero padded (77 bits -> 10 bytes)."""
    b = np.asarray(bits, dtype=np.uint8)
    pad = (-len(b)) % 8
    return np.packbits(np.concatenate([b, np.zeros(pad, dtype=np.uint8)])).tobytes()


@dataclasses.dataclass
class Ping:
    """n_frames consecutive copies of one MSK144 frame starting at sample `start`."""
    msg77: np.ndarray
    start: int
    n_frames: int
    freq_hz: float       # carrier (audio: around 1500; IQ: around 0)
    snr_db: float
    phase: float = 0.0


def _ping_baseband(p: Ping) -> np.ndarray:
    frame = modulate_frame(frame_bits(encode_message(p.msg77)))
    return np.tile(frame, p.n_frames)


def synth_audio(n_samples: int, pings: Iterable[Ping], noise_sigma: float, rng: np.random.Generator) -> np.ndarray:
    """int16 mono 12 kHz.  SNR is in 2500 Hz: 10log10((A^2/2)/(sigma^2*2500/6000))."""
    x = rng.normal(0.0, noise_sigma, size=n_samples) if noise_sigma > 0 else np.zeros(n_samples)
    for p in pings:
        bb = _ping_baseband(p)
        amp = np.sqrt(2.0 * (noise_sigma ** 2 if noise_sigma > 0 else 1.0) * (2500.0 / 6000.0) * 10.0 ** (p.snr_db / 10.0))
        n0 = p.start
        n1 = min(n_samples, n0 + len(bb))
        if n1 <= n0:
            continue
        n = np.arange(n0, n1)
        carrier = np.exp(1j * (2 * np.pi * p.freq_hz * n / SAMPLE_RATE + p.phase))
        x[n0:n1] += amp * np.real(bb[:n1 - n0] * carrier)
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def synth_iq(n_samples: int, pings: Iterable[Ping], noise_sigma: float, rng: np.random.Generator) -> np.ndarray:
    """int8 interleaved I,Q at 12 kHz, shape (2*n_samples,).  noise_sigma is per rail;
    SNR in 2500 Hz: 10log10(A^2/(2 sigma^2 * 2500/12000))."""
    if noise_sigma > 0:
        x = rng.normal(0.0, noise_sigma, size=n_samples) + 1j * rng.normal(0.0, noise_sigma, size=n_samples)
    else:
        x = np.zeros(n_samples, dtype=np.complex128)
    for p in pings:
        bb = _ping_baseband(p)
        amp = np.sqrt(2.0 * (noise_sigma ** 2 if noise_sigma > 0 else 1.0) * (2500.0 / 12000.0) * 10.0 ** (p.snr_db / 10.0))
        n0 = p.start
        n1 = min(n_samples, n0 + len(bb))
        if n1 <= n0:
            continue
        n = np.arange(n0, n1)
        carrier = np.exp(1j * (2 * np.pi * p.freq_hz * n / SAMPLE_RATE + p.phase))
        x[n0:n1] += amp * bb[:n1 - n0] * carrier
    out = np.empty(2 * n_samples, dtype=np.int8)
    out[0::2] = np.clip(np.rint(x.real), -128, 127).astype(np.int8)
    out[1::2] = np.clip(np.rint(x.imag), -128, 127).astype(np.int8)
    return out


# --------------------------------------------------------------------------------------------
# Named stream recipes.  Draw order per stream, from Generator(PCG64(seed)):
#   for each ping: message bits (77), start sample, frame count, frequency offset, phase, SNR pick;
#   then the noise samples.  Seeds: 0x4D534B31 + stream index.
# --------------------------------------------------------------------------------------------
SEED_BASE = 0x4D534B31


def _pings(rng, n_samples, n_pings, center, snr_choices, max_frames=8, span=240.0):
    pings = []
    for _ in range(n_pings):
        msg = random_message(rng)
        n_frames = int(rng.integers(1, max_frames + 1))
        start = int(rng.integers(0, max(1, n_samples - n_frames * FRAME_SAMPLES)))
        df = float(rng.uniform(-span, span))
        phase = float(rng.uniform(0, 2 * np.pi))
        snr = float(snr_choices[int(rng.integers(0, len(snr_choices)))])
        pings.append(Ping(msg, start, n_frames, center + df, snr, phase))
    return pings


def stream_s1(index: int, seconds: float = 30.0, n_pings: int = 6, span: float = 240.0):
    """Functional audio stream: sigma=1000 LSB, SNR in {-4,0,+6} dB.  Returns (int16 samples, pings)."""
    rng = np.random.default_rng(SEED_BASE + index)
    n = int(seconds * SAMPLE_RATE)
    pings = _pings(rng, n, n_pings, 1500.0, (-4.0, 0.0, 6.0), span=span)
    return synth_audio(n, pings, 1000.0, rng), pings


def stream_s2(index: int, seconds: float = 10.0, span: float = 240.0):
    """Throughput audio stream: 2 pings per 10 s at 0 dB."""
    rng = np.random.default_rng(SEED_BASE + 0x10000 + index)
    n = int(seconds * SAMPLE_RATE)
    pings = _pings(rng, n, max(1, int(round(2 * seconds / 10.0))), 1500.0, (0.0,), span=span)
    return synth_audio(n, pings, 1000.0, rng), pings


def stream_s3(index: int, seconds: float = 10.0, span: float = 240.0):
    """LDPC-stress IQ stream: sigma=20 LSB per rail, pings at -6..-2 dB around 0 Hz."""
    rng = np.random.default_rng(SEED_BASE + 0x20000 + index)
    n = int(seconds * SAMPLE_RATE)
    pings = _pings(rng, n, max(1, int(round(2 * seconds / 10.0))), 0.0, (-6.0, -4.0, -2.0), span=span)
    return synth_iq(n, pings, 20.0, rng), pings


def windows_of(stream: np.ndarray, read_mode: int = 1):
    """Split a stream into the decoder's 50 %-overlap windows (main.cu:271-294, 337-359)."""
    per = 1 if read_mode == 1 else 2
    win = WINDOW_SAMPLES * per
    hop = win // 2
    out = []
    s = 0
    while s + win <= len(stream):
        out.append(stream[s:s + win])
        s += hop
    return out


def iq_low_snr_batch(n_channels: int, seed: int = 5):
    """BASELINE configs[4] workload: one int8 I/Q window per channel (S3 style): complex AWGN sigma = 20 LSB per
    rail; every 4th channel carries one ping of 3-6 frames at -6..-2 dB, carrier within +-240 Hz of 0 Hz.
    Returns (windows int8 [n][2*5184], {channel: packed 10-byte payload})."""
    rng = np.random.default_rng(seed)
    wins = np.empty((n_channels, 2 * WINDOW_SAMPLES), dtype=np.int8)
    truth = {}
    for ch in range(n_channels):
        pings = []
        if ch % 4 == 0:
            msg = random_message(rng)
            pings = [Ping(msg, int(rng.integers(0, 1500)), int(rng.integers(3, 7)), float(rng.uniform(-240, 240)), float(rng.uniform(-6, -2)),
                          float(rng.uniform(0, 6.28)))]
            truth[ch] = bytes(np.packbits(np.concatenate([msg, np.zeros(3, np.uint8)])))
        wins[ch] = synth_iq(WINDOW_SAMPLES, pings, 20.0, rng)
    return wins, truth
