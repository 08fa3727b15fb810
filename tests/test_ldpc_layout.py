"""csrc/ldpc_layout.h (the searched lane layout of the BP messages) checked on the CPU: the tables are bijections and, under the
LDS bank rules of MI355X_MICROARCH.md (ds_read_b32: two groups of 32 lanes, 32 banks of 4 bytes, N distinct addresses on one bank
= N cycles, equal addresses broadcast), the gathers have exactly the extra cycles the header quotes.

The model below lists the FULL address of EVERY read the kernel issues, as ldpc.hip's make_edge_tables() derives them - including
the constant-1.0 read of a degree-10 check in its empty round.  (Round 2's model left that read out; the PMC counters showed 20
conflict cycles per iteration where the model said 11, and with the read modelled the round-2 tables score 9 + 11 = 20.)"""
import os
import re

import numpy as np

from msk144cudecoder_amd import protocol as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = open(os.path.join(ROOT, "msk144cudecoder_amd", "csrc", "ldpc_layout.h")).read()


def _table(name):
    m = re.search(name + r"(?:\[\d+\])+\s*=\s*\{(.*?)\};", HDR, re.S)
    return [int(x) for x in re.findall(r"\d+", m.group(1))]


def _const(name):
    return int(re.search(name + r" = (\d+);", HDR).group(1))


ONES = _const("kOnesBase")
TILE = _const("kTileCells")
ROWBASE = _table("kRowBase")
BITS = np.array(_table("kBitOfLane")).reshape(2, 64)
SWAP = np.array(_table("kSwapFirstEdges"))
LOC = np.array(_table("kLaneOfCheck"))
ROUND = np.array(_table("kRoundOfSlot")).reshape(38, 11)
ONE_CELL = _table("kOneCellOfCheck")
ROWS = [[n for n in r if n >= 0] for r in P.CHECK_BITS]
EDGES = [[] for _ in range(128)]
for c, r in enumerate(ROWS):
    for j, n in enumerate(r):
        EDGES[n].append((j, c))


def extra_cycles(addresses):
    banks = {}
    for a in addresses:
        banks.setdefault(a % 32, set()).add(a)
    return max(len(v) for v in banks.values()) - 1


def edge_of(bits, swap, h, lane, i):
    n = int(bits[h][lane])
    k = 1 - i if (i < 2 and swap[n]) else i
    return EDGES[n][k]


def backward_accesses(bits, swap, loc, rounds, rowbase):
    """The six gather loads of the bit lanes x two 32-lane groups: cell = kRowBase[round] + lane of the check."""
    acc = []
    for h in range(2):
        for i in range(3):
            for g in range(2):
                adr = []
                for lane in range(32 * g, 32 * g + 32):
                    j, c = edge_of(bits, swap, h, lane, i)
                    adr.append(rowbase[rounds[c][j]] + int(loc[c]))
                acc.append(adr)
    return acc


def forward_accesses(bits, swap, loc, rounds, one_cell):
    """The eleven gather loads of the check lanes x two 32-lane groups (lanes >= 38 are masked off): a real edge's bit-major cell
    (3h + i)*64 + lane of the bit, or the check's 1.0 cell in the empty round of a degree-10 check."""
    cell = {}
    for h in range(2):
        for lane in range(64):
            for i in range(3):
                j, c = edge_of(bits, swap, h, lane, i)
                cell[(c, j)] = (3 * h + i) * 64 + lane
    assert len(cell) == 384
    acc = {(r, g): [] for r in range(11) for g in range(2)}
    for c, row in enumerate(ROWS):
        for j in range(11):
            acc[(int(rounds[c][j]), int(loc[c] >= 32))].append(cell[(c, j)] if j < len(row) else one_cell[c])
    return list(acc.values())


def total(accesses):
    return sum(extra_cycles(a) for a in accesses)


NATURAL = dict(bits=np.arange(128).reshape(2, 64), swap=np.zeros(128, dtype=int), loc=np.arange(38), rounds=np.tile(np.arange(11), (38, 1)))


def test_tables_are_bijections_and_cells_are_disjoint():
    assert sorted(BITS.reshape(-1).tolist()) == list(range(128))
    assert sorted(LOC.tolist()) == list(range(38))
    assert set(SWAP.tolist()) <= {0, 1} and len(SWAP) == 128
    for c in range(38):
        assert sorted(ROUND[c].tolist()) == list(range(11))   # every check visits every round once (one is empty for degree 10)
    assert ONES == 384 and ONES % 32 == 0                      # forward tile = cells 0..383, then one 1.0 cell per bank
    rows = sorted(ROWBASE)
    assert rows[0] >= ONES + 32 and all(b - a >= 38 for a, b in zip(rows, rows[1:])) and rows[-1] + 38 <= TILE
    cells = {ROWBASE[int(ROUND[c][j])] + int(LOC[c]) for c, r in enumerate(ROWS) for j in range(len(r))}
    assert len(cells) == 384                                   # one backward cell per Tanner-graph edge
    for c, r in enumerate(ROWS):
        if len(r) == 10:
            assert ONES <= ONE_CELL[c] < ONES + 32
    assert TILE * 4 * 32 <= 160 * 1024                         # 32 resident waves per CU fit the LDS


def test_forward_gather_counts_every_read():
    fwd = total(forward_accesses(BITS, SWAP, LOC, ROUND, ONE_CELL))
    single = total(forward_accesses(BITS, SWAP, LOC, ROUND, [ONES] * 38))
    natural = total(forward_accesses(NATURAL["bits"], NATURAL["swap"], NATURAL["loc"], NATURAL["rounds"], [ONES] * 38))
    quoted = re.search(r"Forward gather: (\d+) extra LDS cycles per iteration \(natural layout, rounds = row order, one 1.0 cell: (\d+); "
                       r"these tables with a single 1.0 cell on bank 0: (\d+)\)", HDR)
    assert quoted, "header comment changed"
    assert (fwd, natural, single) == tuple(int(x) for x in quoted.groups())
    assert fwd == 0
    assert single > 0 and natural > 0      # the model sees the 1.0 reads: parked on one bank they do collide with real reads


def test_backward_gather_matches_the_quoted_conflicts():
    acc = backward_accesses(BITS, SWAP, LOC, ROUND, ROWBASE)
    bwd = total(acc)
    natural = total(backward_accesses(NATURAL["bits"], NATURAL["swap"], NATURAL["loc"], NATURAL["rounds"], [ONES + 8 + 38 * r for r in range(11)]))
    quoted = re.search(r"Backward gather: (\d+) extra LDS cycles per iteration \(natural layout, bit n in lane n % 64, rows 38 cells apart: (\d+)\)", HDR)
    assert quoted, "header comment changed"
    assert (bwd, natural) == tuple(int(x) for x in quoted.groups())
    assert bwd <= 4 and max(extra_cycles(a) for a in acc) <= 1 and natural > bwd


def test_conflict_total_is_what_the_counters_are_checked_against():
    """profiles/counters.json: SQ_LDS_BANK_CONFLICT / (SQ_INSTS_LDS / 34) must come out at this total (tools/make_counters_json.py)."""
    tot = total(forward_accesses(BITS, SWAP, LOC, ROUND, ONE_CELL)) + total(backward_accesses(BITS, SWAP, LOC, ROUND, ROWBASE))
    assert tot <= 11
