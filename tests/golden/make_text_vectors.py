"""Prints the literal 77-bit <-> text pairs pasted into tests/test_host.py (TEXT_VECTORS).

NOT WSJT-X output (WSJT-X is absent from this environment and from the reference tree: the text layer stays "parity unpinned").
Every vector is assembled from the PUBLISHED field definitions of the 77-bit messages (Franke, Somerville, Taylor: "The FT4 and FT8
Communication Protocols", QEX Jul/Aug 2020, tables 1-3 and the WSJT-X User Guide) with plain integer arithmetic written out below -
independently of tests/pack77.py and of host/unpack77.cpp - and then frozen as text constants, so that neither the packer nor the
unpacker can drift without a test noticing.

    python tests/golden/make_text_vectors.py
"""
NTOKENS, MAX22 = 2063592, 4194304
A1, A2, A3, A4 = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ", "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ", "0123456789", " ABCDEFGHIJKLMNOPQRSTUVWXYZ"
A38 = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ/"
A42 = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ+-./?"


def bits(value, n):
    assert 0 <= value < (1 << n)
    return "".join(str((value >> (n - 1 - i)) & 1) for i in range(n))


def std_call(call):
    """Standard callsign: up to six characters with the digit in third place ("K1ABC" -> " K1ABC"); 3DA0xxx travels as 3D0xxx, 3Xxxx
    (Guinea) as Qxxx."""
    if call.startswith("3DA0"):
        call = "3D0" + call[4:]
    elif call.startswith("3X") and call[2].isalpha():
        call = "Q" + call[2:]
    c = call if call[2].isdigit() and len(call) <= 6 else " " + call
    c = c.ljust(6)
    assert len(c) == 6 and c[2].isdigit()
    n = A1.index(c[0])
    n = n * 36 + A2.index(c[1])
    n = n * 10 + A3.index(c[2])
    for ch in c[3:]:
        n = n * 27 + A4.index(ch)
    return NTOKENS + MAX22 + n


def hash_call(call, m):
    n8 = 0
    for ch in call.ljust(11):
        n8 = n8 * 38 + A38.index(ch)
    return ((47055833459 * n8) & ((1 << 64) - 1)) >> (64 - m)


def cq_number(n):          # "CQ 000" .. "CQ 999"
    return 3 + n


def cq_letters(s):         # "CQ A" .. "CQ ZZZZ": one to four letters, right-justified, base 27
    v = 0
    for ch in s.rjust(4):
        v = v * 27 + A4.index(ch)
    return 1003 + v


def grid4(g):
    return (("ABCDEFGHIJKLMNOPQR".index(g[0]) * 18 + "ABCDEFGHIJKLMNOPQR".index(g[1])) * 10 + int(g[2])) * 10 + int(g[3])


def report(db):            # -30 .. +99 as the 15-bit field's report range
    return 32400 + 35 + db


def grid6(g):
    L, X = "ABCDEFGHIJKLMNOPQR", "ABCDEFGHIJKLMNOPQRSTUVWX"
    return ((((L.index(g[0]) * 18 + L.index(g[1])) * 10 + int(g[2])) * 10 + int(g[3])) * 24 + X.index(g[4])) * 24 + X.index(g[5])


def standard(n28a, pa, n28b, pb, r, g15, i3):
    return bits(n28a, 28) + str(pa) + bits(n28b, 28) + str(pb) + str(r) + bits(g15, 15) + bits(i3, 3)


def nonstandard(hashed, call11, flip, rpt, cq):
    n58 = 0
    for ch in call11.rjust(11):
        n58 = n58 * 38 + A38.index(ch)
    return bits(hash_call(hashed, 12), 12) + bits(n58, 58) + str(flip) + bits(rpt, 2) + str(cq) + bits(4, 3)


def free_text(s):
    n = 0
    for ch in s.rjust(13):
        n = n * 42 + A42.index(ch)
    return bits(n, 71) + bits(0, 3) + bits(0, 3)


def telemetry(hex18):
    return bits(int(hex18, 16), 71) + bits(5, 3) + bits(0, 3)


def euvhf(call_h12, call_h22, r, rst, serial, g6):
    return bits(hash_call(call_h12, 12), 12) + bits(hash_call(call_h22, 22), 22) + str(r) + bits(rst - 52, 3) + bits(serial, 11) + bits(grid6(g6), 25) + bits(5, 3)


K, W = std_call("K1ABC"), std_call("W9XYZ")
VECTORS = [
    # (what must have been heard before, 77 bits, text)
    ("", standard(2, 0, K, 0, 0, grid4("FN42"), 1), "CQ K1ABC FN42"),
    ("", standard(K, 0, W, 0, 0, grid4("EN37"), 1), "K1ABC W9XYZ EN37"),
    ("", standard(W, 0, K, 0, 0, report(-11), 1), "W9XYZ K1ABC -11"),
    ("", standard(K, 0, W, 0, 1, report(-9), 1), "K1ABC W9XYZ R-09"),
    ("", standard(K, 0, W, 0, 0, report(7), 1), "K1ABC W9XYZ +07"),
    ("", standard(W, 0, K, 0, 0, 32400 + 2, 1), "W9XYZ K1ABC RRR"),
    ("", standard(K, 0, W, 0, 0, 32400 + 3, 1), "K1ABC W9XYZ RR73"),
    ("", standard(K, 0, W, 0, 0, 32400 + 4, 1), "K1ABC W9XYZ 73"),
    ("", standard(K, 0, W, 0, 0, 32400 + 1, 1), "K1ABC W9XYZ"),
    ("", standard(0, 0, K, 0, 0, 32400 + 1, 1), "DE K1ABC"),
    ("", standard(1, 0, std_call("PA9XYZ"), 0, 0, grid4("JO22"), 1), "QRZ PA9XYZ JO22"),
    ("", standard(cq_number(123), 0, std_call("G4ABC"), 0, 0, grid4("IO91"), 1), "CQ 123 G4ABC IO91"),
    ("", standard(cq_number(7), 0, std_call("G4ABC"), 0, 0, grid4("IO91"), 1), "CQ 007 G4ABC IO91"),
    ("", standard(cq_letters("DX"), 0, std_call("RA9YER"), 0, 0, grid4("MO05"), 1), "CQ DX RA9YER MO05"),
    ("", standard(cq_letters("TEST"), 0, K, 0, 0, grid4("FN42"), 1), "CQ TEST K1ABC FN42"),
    ("", standard(cq_letters("A"), 0, K, 0, 0, grid4("FN42"), 1), "CQ A K1ABC FN42"),
    ("", standard(K, 1, W, 0, 0, grid4("EN37"), 1), "K1ABC/R W9XYZ EN37"),
    ("", standard(W, 0, K, 1, 1, grid4("FN42"), 1), "W9XYZ K1ABC/R R FN42"),
    ("", standard(std_call("G4ABC"), 1, std_call("PA9XYZ"), 0, 0, grid4("JO22"), 2), "G4ABC/P PA9XYZ JO22"),
    ("", standard(std_call("G4ABC"), 0, std_call("PA9XYZ"), 1, 1, grid4("JO22"), 2), "G4ABC PA9XYZ/P R JO22"),
    ("", standard(2, 0, std_call("3DA0XYZ"), 0, 0, grid4("KG53"), 1), "CQ 3DA0XYZ KG53"),
    ("", standard(2, 0, std_call("3XY1A"), 0, 0, grid4("IJ39"), 1), "CQ 3XY1A IJ39"),
    ("", standard(K, 0, std_call("KH7Z"), 0, 0, grid4("AJ10"), 1), "K1ABC KH7Z AJ10"),
    ("", standard(2, 0, std_call("W9XYZ"), 0, 0, grid4("RR99"), 1), "CQ W9XYZ RR99"),
    ("", standard(2, 0, std_call("K1ABC"), 0, 0, grid4("AA00"), 1), "CQ K1ABC AA00"),
    # hashed calls: 22-bit hash in a 28-bit field; 12-bit hash + 58-bit call in type 4; both hashes in type 5
    ("", standard(NTOKENS + hash_call("PJ4/K1ABC", 22), 0, W, 0, 0, report(-3), 1), "<...> W9XYZ -03"),
    ("PJ4/K1ABC", standard(NTOKENS + hash_call("PJ4/K1ABC", 22), 0, W, 0, 0, report(-3), 1), "<PJ4/K1ABC> W9XYZ -03"),
    ("", nonstandard("W9XYZ", "PJ4/K1ABC", 0, 0, 0), "<...> PJ4/K1ABC"),
    ("W9XYZ", nonstandard("W9XYZ", "PJ4/K1ABC", 0, 0, 0), "<W9XYZ> PJ4/K1ABC"),
    ("W9XYZ", nonstandard("W9XYZ", "PJ4/K1ABC", 1, 2, 0), "PJ4/K1ABC <W9XYZ> RR73"),
    ("W9XYZ", nonstandard("W9XYZ", "YW18FIFA", 0, 1, 0), "<W9XYZ> YW18FIFA RRR"),
    ("W9XYZ", nonstandard("W9XYZ", "YW18FIFA", 1, 3, 0), "YW18FIFA <W9XYZ> 73"),
    ("", nonstandard("W9XYZ", "YW18FIFA", 0, 0, 1), "CQ YW18FIFA"),
    ("", nonstandard("W9XYZ", "PJ4/K1ABC", 0, 0, 1), "CQ PJ4/K1ABC"),
    ("G4ABC,PA9XYZ", euvhf("G4ABC", "PA9XYZ", 1, 57, 7, "JO22DB"), "<G4ABC> <PA9XYZ> R 570007 JO22DB"),
    ("G4ABC,PA9XYZ", euvhf("PA9XYZ", "G4ABC", 0, 59, 1234, "IO91NP"), "<PA9XYZ> <G4ABC> 591234 IO91NP"),
    ("", euvhf("PA9XYZ", "G4ABC", 0, 52, 1, "AA00AA"), "<...> <...> 520001 AA00AA"),
    # free text (13 characters base 42, right-justified) and telemetry (18 hex digits, leading zeros dropped)
    ("", free_text("TNX BOB 73 GL"), "TNX BOB 73 GL"),
    ("", free_text("HELLO WORLD"), "HELLO WORLD"),
    ("", free_text("1/2+3-4.5?"), "1/2+3-4.5?"),
    ("", free_text("A"), "A"),
    ("", telemetry("123456789ABCDEF012"), "123456789ABCDEF012"),
    ("", telemetry("00000000000000ABCD"), "ABCD"),
    ("", telemetry("7FFFFFFFFFFFFFFFFF"), "7FFFFFFFFFFFFFFFFF"),
]

if __name__ == "__main__":
    print("TEXT_VECTORS = [")
    for heard, b, text in VECTORS:
        assert len(b) == 77 and set(b) <= {"0", "1"}
        print(f'    ("{heard}", "{b}", "{text}"),')
    print("]")
