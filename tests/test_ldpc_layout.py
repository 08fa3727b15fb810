"""csrc/ldpc_layout.h (the searched lane layout of the BP messages) checked on the CPU: the tables are bijections and, under the
LDS bank rules of MI355X_MICROARCH.md (ds_read_b32: two groups of 32 lanes, 32 banks of 4 bytes, N distinct addresses on one bank
= N cycles), the forward gather of the check lanes is conflict-free in every round (the rounds are a proper edge colouring of
(check, bit lane mod 32)) and the backward gather of the bit lanes has exactly the extra cycles the header quotes."""
import os
import re

import numpy as np

from msk144cudecoder_amd import protocol as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = open(os.path.join(ROOT, "msk144cudecoder_amd", "csrc", "ldpc_layout.h")).read()


def _table(name):
    m = re.search(name + r"(?:\[\d+\])+\s*=\s*\{(.*?)\};", HDR, re.S)
    return [int(x) for x in re.findall(r"\d+", m.group(1))]


S = int(re.search(r"kTileRowStride = (\d+);", HDR).group(1))
BITS = np.array(_table("kBitOfLane")).reshape(2, 64)
SWAP = np.array(_table("kSwapFirstEdges"))
LOC = np.array(_table("kLaneOfCheck"))
ROUND = np.array(_table("kRoundOfSlot")).reshape(38, 11)
ROWS = [[n for n in r if n >= 0] for r in P.CHECK_BITS]
EDGES = [[] for _ in range(128)]
for c, r in enumerate(ROWS):
    for j, n in enumerate(r):
        EDGES[n].append((j, c))


def backward_conflicts(bits, swap, loc, rounds, stride):
    """Extra LDS cycles of the six gather loads of the bit lanes (cell = round * stride + lane of the check)."""
    total, worst = 0, 0
    for h in range(2):
        for i in range(3):
            for g in range(2):
                banks = []
                for lane in range(32 * g, 32 * g + 32):
                    n = bits[h][lane]
                    k = 1 - i if (i < 2 and swap[n]) else i
                    j, c = EDGES[n][k]
                    banks.append((rounds[c][j] * stride + loc[c]) % 32)
                m = int(np.bincount(banks, minlength=32).max())
                total += m - 1
                worst = max(worst, m)
    return total, worst


def forward_conflicts(bits, loc, rounds):
    """Extra LDS cycles of the eleven gather loads of the check lanes (cell = (3h + i) * 64 + lane of the bit: bank = lane % 32)."""
    lane_of_bit = {int(bits[h][l]): l for h in range(2) for l in range(64)}
    total = 0
    for r in range(11):
        for g in range(2):
            cells = {}
            for c, row in enumerate(ROWS):
                if (loc[c] >= 32) != bool(g):
                    continue
                for j, n in enumerate(row):
                    if rounds[c][j] == r:
                        cells.setdefault(lane_of_bit[n] % 32, set()).add(n)
            if cells:
                total += max(len(v) for v in cells.values()) - 1
    return total


def test_tables_are_bijections():
    assert sorted(BITS.reshape(-1).tolist()) == list(range(128))
    assert sorted(LOC.tolist()) == list(range(38))
    assert set(SWAP.tolist()) <= {0, 1} and len(SWAP) == 128
    assert S >= 38                                     # rows of 38 check lanes do not overlap
    for c in range(38):
        assert sorted(ROUND[c].tolist()) == list(range(11))   # every check visits every round once (one is empty for degree 10)
    cells = {int(ROUND[c][j]) * S + int(LOC[c]) for c, r in enumerate(ROWS) for j in range(len(r))}
    assert len(cells) == 384                            # one backward cell per Tanner-graph edge


def test_forward_gather_is_conflict_free():
    assert forward_conflicts(BITS, LOC, ROUND) == 0
    natural = forward_conflicts(np.arange(128).reshape(2, 64), np.arange(38), np.tile(np.arange(11), (38, 1)))
    quoted = re.search(r"Forward gather: (\d+) extra LDS cycles per iteration \(rounds = row order on the natural layout: (\d+)\)", HDR)
    assert quoted, "header comment changed"
    assert (0, natural) == (int(quoted.group(1)), int(quoted.group(2))) and natural > 0


def test_backward_gather_matches_the_quoted_conflicts():
    total, worst = backward_conflicts(BITS, SWAP, LOC, ROUND, S)
    quoted = re.search(r"Backward gather: (\d+) extra LDS cycles per iteration \(natural layout, bit n in lane n % 64, rounds = row order: (\d+)\)", HDR)
    assert quoted, "header comment changed"
    assert total == int(quoted.group(1)) and worst <= 3
    natural = backward_conflicts(np.arange(128).reshape(2, 64), np.zeros(128, dtype=int), np.arange(38), np.tile(np.arange(11), (38, 1)), S)
    assert natural[0] == int(quoted.group(2)) and natural[0] > total
