"""csrc/ldpc_layout.h (the annealed lane layout of the BP message tile) checked on the CPU: the tables are bijections and, under
the LDS bank rules of MI355X_MICROARCH.md (ds_read_b32 / ds_write_b32: two groups of 32 lanes, 32 banks of 4 bytes, one extra
cycle per extra distinct address on a bank), no edge-side instruction is worse than 2-way conflicted and the totals are the ones
the header quotes; the check side (consecutive lanes on consecutive cells) is conflict-free for any stride."""
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
ROWS = [[n for n in r if n >= 0] for r in P.CHECK_BITS]
EDGES = [[] for _ in range(128)]
for c, r in enumerate(ROWS):
    for j, n in enumerate(r):
        EDGES[n].append((j, c))


def edge_conflicts(bits, swap, loc, stride):
    total, worst = 0, 0
    for h in range(2):
        for i in range(3):
            for g in range(2):
                banks = []
                for lane in range(32 * g, 32 * g + 32):
                    n = bits[h][lane]
                    k = 1 - i if (i < 2 and swap[n]) else i
                    j, c = EDGES[n][k]
                    banks.append((j * stride + loc[c]) % 32)
                m = int(np.bincount(banks, minlength=32).max())
                total += m - 1
                worst = max(worst, m)
    return total, worst


def test_tables_are_bijections():
    assert sorted(BITS.reshape(-1).tolist()) == list(range(128))
    assert sorted(LOC.tolist()) == list(range(38))
    assert set(SWAP.tolist()) <= {0, 1} and len(SWAP) == 128
    assert S >= 38                                     # rows of 38 check lanes do not overlap
    cells = {j * S + LOC[c] for c, r in enumerate(ROWS) for j in range(len(r))}
    assert len(cells) == 384                            # one cell per Tanner-graph edge


def test_edge_side_is_at_most_two_way_conflicted():
    total, worst = edge_conflicts(BITS, SWAP, LOC, S)
    quoted = re.search(r"direction: (\d+) \(worst instruction (\d+)-way\); the natural layout\s*//\s*\(bit n in lane n % 64, stride 40\) has (\d+) \((\d+)-way\)", HDR)
    assert quoted, "header comment changed"
    assert (total, worst) == (int(quoted.group(1)), int(quoted.group(2)))
    assert worst <= 2 and total <= 9
    natural = edge_conflicts(np.arange(128).reshape(2, 64), np.zeros(128, dtype=int), np.arange(38), 40)
    assert natural == (int(quoted.group(3)), int(quoted.group(4))) and natural[0] >= 2 * total


def test_check_side_is_conflict_free():
    for j in range(11):
        cells = [j * S + lane for lane in range(38)]    # check lane l walks column cell j*S + l
        for group in (cells[:32], cells[32:]):
            assert len({c % 32 for c in group}) == len(group)
