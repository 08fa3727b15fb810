"""Test-side packer for the 77-bit message families the text layer handles, written from the
protocol description (the inverse of msk144cudecoder_amd/host/unpack77.cpp; both unpinned against
WSJT-X, whose source is absent)."""
import numpy as np

A1 = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ"
A2 = "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ"
A3 = "0123456789"
A4 = " ABCDEFGHIJKLMNOPQRSTUVWXYZ"
A38 = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ/"
A42 = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ+-./?"
NTOKENS, MAX22, MAXGRID4 = 2063592, 4194304, 32400


def bits_of(value, n):
    return [(value >> (n - 1 - i)) & 1 for i in range(n)]


def hash_call(call, m):
    c = call.ljust(11)[:11]
    n8 = 0
    for ch in c:
        n8 = 38 * n8 + A38.index(ch)
    return ((47055833459 * n8) & 0xFFFFFFFFFFFFFFFF) >> (64 - m)


def pack28(token):
    if token == "DE":
        return 0
    if token == "QRZ":
        return 1
    if token == "CQ":
        return 2
    if token.startswith("CQ ") and token[3:].isdigit():
        return 3 + int(token[3:])
    if token.startswith("CQ "):
        s = token[3:].rjust(4)
        n = 0
        for ch in s:
            n = n * 27 + A4.index(ch)
        return 1003 + n
    if token.startswith("<"):
        return NTOKENS + hash_call(token.strip("<>"), 22)
    call = token
    if call.startswith("3DA0"):
        call = "3D0" + call[4:]
    elif call.startswith("3X") and call[2].isalpha():
        call = "Q" + call[2:]
    # right-align so that the digit sits in position 3
    c6 = call if (len(call) > 2 and call[2].isdigit()) else " " + call
    c6 = c6.ljust(6)
    n = A1.index(c6[0])
    n = n * 36 + A2.index(c6[1])
    n = n * 10 + A3.index(c6[2])
    n = n * 27 + A4.index(c6[3])
    n = n * 27 + A4.index(c6[4])
    n = n * 27 + A4.index(c6[5])
    return NTOKENS + MAX22 + n


def pack_standard(c1, c2, extra, i3=1, p1=0, p2=0):
    ir = 0
    if extra.startswith("R ") or (extra.startswith("R") and len(extra) > 1 and extra[1] in "+-"):
        ir = 1
        extra = extra[2:] if extra.startswith("R ") else extra[1:]
    if extra == "":
        g = MAXGRID4 + 1
    elif extra == "RRR":
        g = MAXGRID4 + 2
    elif extra == "RR73":
        g = MAXGRID4 + 3
    elif extra == "73":
        g = MAXGRID4 + 4
    elif extra[0] in "+-":
        snr = int(extra)
        g = MAXGRID4 + 35 + (snr if snr >= -30 else snr + 101)
        if snr < 0:
            g = MAXGRID4 + 35 + snr
    else:
        g = ((ord(extra[0]) - 65) * 18 + (ord(extra[1]) - 65)) * 100 + int(extra[2]) * 10 + int(extra[3])
    b = bits_of(pack28(c1), 28) + [p1] + bits_of(pack28(c2), 28) + [p2] + [ir] + bits_of(g, 15) + bits_of(i3, 3)
    return np.array(b, dtype=np.uint8)


def pack_free_text(text):
    t = text.rjust(13)[:13]
    n = 0
    for ch in t:
        n = n * 42 + A42.index(ch)
    return np.array(bits_of(n, 71) + bits_of(0, 3) + bits_of(0, 3), dtype=np.uint8)


def pack_telemetry(hexstr):
    n = int(hexstr, 16)
    return np.array(bits_of(n, 71) + bits_of(5, 3) + bits_of(0, 3), dtype=np.uint8)


def pack_nonstandard(hashed_call, full_call, flip=0, rpt=0, cq=0):
    n58 = 0
    for ch in full_call.rjust(11):
        n58 = n58 * 38 + A38.index(ch)
    b = bits_of(hash_call(hashed_call, 12), 12) + bits_of(n58, 58) + [flip] + bits_of(rpt, 2) + [cq] + bits_of(4, 3)
    return np.array(b, dtype=np.uint8)
