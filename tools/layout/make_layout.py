"""Offline generator of msk144cudecoder_amd/csrc/ldpc_layout.h (developer tool, not part of the product).

The search itself is tools/layout/anneal_rounds.cpp (simulated annealing over bit -> lane, first-edge order, check -> lane, the
round of every edge at its check and the bank residue of every backward row); this script turns its result into the header:
it places the backward rows (first-fit, each at its residue mod 32), gives every empty round of a degree-10 check a constant-1.0
cell on a bank no real read of the same (round, 32-lane group) access uses, re-counts every conflict with the full-address
model below and quotes the counts in the header (tests/test_ldpc_layout.py re-derives them from the emitted tables).

LDS traffic of one BP iteration in ldpc.hip (per wave; cells are floats, bank = cell mod 32, a ds_read_b32 is served as two
groups of 32 lanes, N distinct addresses on one bank = N cycles, equal addresses broadcast):
  forward  (bit -> check): lane (h, l) stores the tanh of its edge of instruction i to cell (3h + i)*64 + l (add-TID store, never
           conflicted); check lane L gathers one factor per round r - a real edge's cell, or, in the empty round of a degree-10
           check, one of the 32 constant-1.0 cells kOnesBase + b;
  backward (check -> bit): check lane L stores its round-r product to cell kRowBase[r] + L (add-TID store); lane (h, l) gathers the
           six cells of its edges.

    g++ -O2 -o /tmp/anneal_rounds tools/layout/anneal_rounds.cpp
    /tmp/anneal_rounds tools/layout/edges.txt SEED ITERATIONS > /tmp/layout.txt
    python tools/layout/make_layout.py /tmp/layout.txt > msk144cudecoder_amd/csrc/ldpc_layout.h
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from msk144cudecoder_amd import protocol as P  # noqa: E402

ROWS = [[n for n in r if n >= 0] for r in P.CHECK_BITS]
EDGES = [[] for _ in range(128)]          # bit -> [(slot j, check c)] in ascending check order (the reference's k)
for c, r in enumerate(ROWS):
    for j, n in enumerate(r):
        EDGES[n].append((j, c))
ROUNDS = 11
ROW_CELLS = 38                             # one backward row = one cell per check lane
FWD_CELLS = 2 * 3 * 64                     # bit-major forward tile
ONES_BASE = FWD_CELLS                      # 32 cells of 1.0, cell ONES_BASE + b sits on bank b (384 = 12 * 32)


def write_edges_file(path):
    with open(path, "w") as f:
        for n in range(128):
            f.write(" ".join(f"{j} {c}" for j, c in EDGES[n]) + "\n")


def cycles(addresses):
    """Extra LDS cycles of one 32-lane access: max over banks of the number of DISTINCT addresses, minus one."""
    per_bank = {}
    for a in addresses:
        per_bank.setdefault(a % 32, set()).add(a)
    return max((len(v) for v in per_bank.values()), default=1) - 1


class Layout:
    def __init__(self, bit_at, swap, loc, rounds, row_base, one_cell):
        self.bit_at, self.swap, self.loc, self.rounds, self.row_base, self.one_cell = bit_at, swap, loc, rounds, row_base, one_cell

    def edge_of_instruction(self, pos, i):
        n = self.bit_at[pos]
        k = 1 - i if (i < 2 and self.swap[n]) else i
        return EDGES[n][k]

    def forward_cell(self):
        """(check, slot) -> forward-tile cell its tanh is stored to."""
        cell = {}
        for pos in range(128):
            h, l = divmod(pos, 64)
            for i in range(3):
                j, c = self.edge_of_instruction(pos, i)
                cell[(c, j)] = (3 * h + i) * 64 + l
        return cell

    def forward_reads(self, one_cell=None):
        """Addresses of the eleven gathers of the check lanes: {(round, group): [cell of every active lane]}."""
        one_cell = self.one_cell if one_cell is None else one_cell
        cell = self.forward_cell()
        acc = {(r, g): [] for r in range(ROUNDS) for g in range(2)}
        for c, row in enumerate(ROWS):
            g = int(self.loc[c] >= 32)
            for j in range(ROUNDS):
                r = self.rounds[c][j]
                acc[(r, g)].append(cell[(c, j)] if j < len(row) else one_cell[c])
        return acc

    def forward_conflicts(self, one_cell=None):
        return sum(cycles(a) for a in self.forward_reads(one_cell).values())

    def backward_conflicts(self):
        total = 0
        for h in range(2):
            for i in range(3):
                for g in range(2):
                    adr = []
                    for lane in range(32 * g, 32 * g + 32):
                        j, c = self.edge_of_instruction(h * 64 + lane, i)
                        adr.append(self.row_base[self.rounds[c][j]] + self.loc[c])
                    total += cycles(adr)
        return total


def place_rows(residue, first_free):
    """Row bases at the wanted residues mod 32, rows of ROW_CELLS cells, first-fit: at each step the row that pads least."""
    base, nxt, left = [None] * ROUNDS, first_free, set(range(ROUNDS))
    while left:
        r = min(left, key=lambda q: ((residue[q] - nxt) % 32, q))
        base[r] = nxt + (residue[r] - nxt) % 32
        nxt = base[r] + ROW_CELLS
        left.remove(r)
    return base, nxt


def assign_one_cells(lay):
    """A constant-1.0 cell for the empty round of every degree-10 check: one bank per (round, group) access that none of its real
    reads uses (all empty lanes of the access share the address: a broadcast)."""
    real = lay.forward_reads(one_cell=[None] * 38)
    one = [0] * 38
    for c, row in enumerate(ROWS):
        if len(row) == ROUNDS:
            continue
        key = (lay.rounds[c][ROUNDS - 1], int(lay.loc[c] >= 32))
        used = {a % 32 for a in real[key] if a is not None}
        one[c] = ONES_BASE + min(b for b in range(32) if b not in used)
    return one


def natural_layout():
    rounds = [list(range(ROUNDS)) for _ in range(38)]
    lay = Layout(list(range(128)), [0] * 128, list(range(38)), rounds, [ONES_BASE + 8 + r * ROW_CELLS for r in range(ROUNDS)], [ONES_BASE] * 38)
    return lay


def read_search_result(path):
    rec = {}
    for line in open(path):
        w = line.split()
        if w:
            rec[w[0]] = [int(x) for x in w[1:]] if w[0] != "seed" else w[1:]
    rounds = [rec["rounds"][11 * c:11 * c + 11] for c in range(38)]
    row_base, end = place_rows(rec["rowres"], ONES_BASE + 32)
    lay = Layout(rec["bits"], rec["swap"], rec["loc"], rounds, row_base, None)
    lay.one_cell = assign_one_cells(lay)
    lay.tile_cells = end
    return lay


def emit(lay):
    fwd, bwd = lay.forward_conflicts(), lay.backward_conflicts()
    nat = natural_layout()
    nfwd, nbwd = nat.forward_conflicts(), nat.backward_conflicts()
    single = lay.forward_conflicts(one_cell=[ONES_BASE] * 38)
    out = []
    out.append("// GENERATED by tools/layout/make_layout.py from a result of tools/layout/anneal_rounds.cpp (simulated annealing over the bit -> lane,")
    out.append("// check -> lane, first-edge-order and round assignments and the bank residues of the backward rows); tests/test_ldpc_layout.py")
    out.append("// re-derives the conflict counts from these tables with the LDS bank rules of MI355X_MICROARCH.md (ds_read_b32: two 32-lane")
    out.append("// groups, 32 banks of 4 bytes, N distinct addresses on a bank = N cycles, equal addresses broadcast) over the full addresses of")
    out.append("// EVERY read the kernel issues, the constant-1.0 reads of the degree-10 checks included.")
    out.append("//")
    out.append("// LDS layout of the BP messages (ldpc.hip): lane l owns codeword bits kBitOfLane[0][l] and kBitOfLane[1][l]; check c is")
    out.append("// processed by lane kLaneOfCheck[c]; kSwapFirstEdges[n] = 1: instruction 0 takes bit n's second edge and instruction 1 its")
    out.append("// first - free, their messages are only ever added to each other first ((tov0 + tov1) + tov2).")
    out.append("//   forward (bit -> check): instruction i of half h stores to the bit-major cell (3h + i)*64 + lane (add-TID store, conflict-free);")
    out.append("//     check lane L reads its edge of round r = kRoundOfSlot[c][j] (j = position of the bit in the check's row).  In its empty")
    out.append("//     round (kRoundOfSlot[c][10] of a degree-10 check) it reads the constant 1.0 at kOneCellOfCheck[c]: one of the 32 cells")
    out.append("//     kOnesBase + b, b = a bank no real read of the same (round, 32-lane group) access uses;")
    out.append("//   backward (check -> bit): the product of round r goes to cell kRowBase[r] + L; the bit side gathers its six cells.")
    out.append(f"// Forward gather: {fwd} extra LDS cycles per iteration (natural layout, rounds = row order, one 1.0 cell: {nfwd}; these tables with a single 1.0 cell on bank 0: {single}).")
    out.append(f"// Backward gather: {bwd} extra LDS cycles per iteration (natural layout, bit n in lane n % 64, rows 38 cells apart: {nbwd}).")
    out.append("#pragma once")
    out.append("")
    out.append("#include <cstdint>")
    out.append("")
    out.append("namespace msk144")
    out.append("{")
    out.append("")
    out.append(f"constexpr int kOnesBase = {ONES_BASE};      // 32 cells of 1.0, one per bank")
    out.append(f"constexpr int kTileCells = {lay.tile_cells};     // floats per wave: forward tile, ones, backward rows")
    out.append("constexpr uint16_t kRowBase[11] = {" + ", ".join(str(x) for x in lay.row_base) + "};")
    out.append("constexpr uint8_t kBitOfLane[2][64] = {")
    for h in range(2):
        out.append("    {" + ", ".join(str(x) for x in lay.bit_at[64 * h:64 * h + 64]) + "},")
    out.append("};")
    out.append("constexpr uint8_t kSwapFirstEdges[128] = {" + ", ".join(str(x) for x in lay.swap) + "};")
    out.append("constexpr uint8_t kLaneOfCheck[38] = {" + ", ".join(str(x) for x in lay.loc) + "};")
    out.append("constexpr uint8_t kRoundOfSlot[38][11] = {")
    for c in range(38):
        out.append("    {" + ", ".join(str(x) for x in lay.rounds[c]) + "},")
    out.append("};")
    out.append("constexpr uint16_t kOneCellOfCheck[38] = {" + ", ".join(str(x) for x in lay.one_cell) + "};  // 0: the check has eleven bits")
    out.append("")
    out.append("}  // namespace msk144")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--edges":
        write_edges_file(sys.argv[2])
    else:
        lay = read_search_result(sys.argv[1])
        sys.stderr.write(f"forward {lay.forward_conflicts()} backward {lay.backward_conflicts()} tile {lay.tile_cells} floats\n")
        sys.stdout.write(emit(lay))
