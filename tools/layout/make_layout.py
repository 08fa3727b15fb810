"""Offline generator of msk144cudecoder_amd/csrc/ldpc_layout.h (developer tool, not part of the product).

LDS traffic of one BP iteration in ldpc.hip:
  forward  (bit -> check): lane (h, l) stores tanh of its edge of instruction i into the BIT-major tile cell (3h + i)*64 + l with
           ds_write_addtid_b32 (no address VGPR, always conflict-free); check lane L then GATHERS its column with eleven ds_read_b32,
           one per round r, from per-lane addresses.  Bank of a cell = lane of the bit mod 32, so the gather of round r is
           conflict-free iff the checks of a 32-lane group read 32 different bit-lane residues: the rounds of a group are a proper
           edge colouring of the bipartite multigraph (check, residue) with eleven colours, which exists iff no residue carries
           more than eleven edges of the group (Koenig).
  backward (check -> bit): check lane L stores its round-r product to cell r*S + L (M0-relative ds_write_addtid_b32), lane (h, l)
           gathers the six cells of its edges: this side keeps bank conflicts, minimised here by local search.
Freedom: bit -> (h, lane), first-two-edges order per bit, check -> lane, and the colouring itself (Kempe chains).

    python tools/layout/make_layout.py [seed] [iterations]  > msk144cudecoder_amd/csrc/ldpc_layout.h
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from msk144cudecoder_amd import protocol as P  # noqa: E402

ROWS = [[n for n in r if n >= 0] for r in P.CHECK_BITS]
EDGES = [[] for _ in range(128)]          # bit -> [(slot j, check c)] in ascending check order (the reference's k)
for c, r in enumerate(ROWS):
    for j, n in enumerate(r):
        EDGES[n].append((j, c))
S = 38                                     # backward tile row stride
ROUNDS = 11


def colour_group(checks, bank_of_edge):
    """Proper edge colouring (rounds) of the bipartite multigraph check x bank for the listed checks.
    bank_of_edge[(c, j)] -> bank.  Returns {(c, j): round} or None if a bank has more than ROUNDS edges."""
    at_check = {c: [None] * ROUNDS for c in checks}     # colour -> edge
    at_bank = {}
    edges = [(c, j) for c in checks for j in range(len(ROWS[c]))]
    deg = {}
    for e in edges:
        deg[bank_of_edge[e]] = deg.get(bank_of_edge[e], 0) + 1
    if max(deg.values()) > ROUNDS:
        return None
    colour = {}
    for e in edges:
        c, b = e[0], bank_of_edge[e]
        tb = at_bank.setdefault(b, [None] * ROUNDS)
        tc = at_check[c]
        fa = next(k for k in range(ROUNDS) if tc[k] is None)     # free at the check
        fb = next(k for k in range(ROUNDS) if tb[k] is None)     # free at the bank
        if tb[fa] is not None:
            # alternating path from the bank along colours fa / fb: swap them on the path
            path, node_is_bank, node, col = [], True, b, fa
            while True:
                tab = at_bank[node] if node_is_bank else at_check[node]
                nxt = tab[col]
                if nxt is None:
                    break
                path.append(nxt)
                node = nxt[0] if node_is_bank else bank_of_edge[nxt]
                node_is_bank = not node_is_bank
                col = fb if col == fa else fa
            for pe in path:                                   # remove, then re-insert with swapped colours
                k = colour[pe]
                at_check[pe[0]][k] = None
                at_bank[bank_of_edge[pe]][k] = None
            for pe in path:
                k = fb if colour[pe] == fa else fa
                colour[pe] = k
                at_check[pe[0]][k] = pe
                at_bank[bank_of_edge[pe]][k] = pe
        assert tc[fa] is None and tb[fa] is None
        colour[e] = fa
        tc[fa] = e
        tb[fa] = e
    return colour


class Layout:
    def __init__(self, rng):
        self.rng = rng
        self.bit_at = list(range(128))
        rng.shuffle(self.bit_at)
        self.swap = [rng.randrange(2) for _ in range(128)]
        self.loc = list(range(38))
        rng.shuffle(self.loc)
        self.round = None

    def pos_of_bit(self):
        pos = [0] * 128
        for p, n in enumerate(self.bit_at):
            pos[n] = p
        return pos

    def colour(self):
        pos = self.pos_of_bit()
        bank = {(c, j): (pos[n] % 64) % 32 for c, r in enumerate(ROWS) for j, n in enumerate(r)}
        g0 = [c for c in range(38) if self.loc[c] < 32]
        g1 = [c for c in range(38) if self.loc[c] >= 32]
        a, b = colour_group(g0, bank), colour_group(g1, bank)
        if a is None or b is None:
            return False
        a.update(b)
        self.round = a
        return True

    def forward_conflicts(self):
        pos = self.pos_of_bit()
        extra = 0
        for r in range(ROUNDS):
            for grp in (0, 1):
                seen = {}
                for c in range(38):
                    if (self.loc[c] >= 32) != bool(grp):
                        continue
                    for j, n in enumerate(ROWS[c]):
                        if self.round[(c, j)] == r:
                            b = (pos[n] % 64) % 32
                            seen.setdefault(b, set()).add(pos[n])
                if seen:
                    extra += max(len(v) for v in seen.values()) - 1
        return extra

    def backward_cost(self):
        """(conflicted cycles, colliding pairs) of the six gather instructions x two groups."""
        cyc = pairs = 0
        for h in range(2):
            for i in range(3):
                for g in range(2):
                    cnt = [0] * 32
                    for lane in range(32 * g, 32 * g + 32):
                        n = self.bit_at[h * 64 + lane]
                        k = 1 - i if (i < 2 and self.swap[n]) else i
                        j, c = EDGES[n][k]
                        cnt[(self.round[(c, j)] * S + self.loc[c]) % 32] += 1
                    cyc += max(cnt) - 1
                    pairs += sum(x * (x - 1) // 2 for x in cnt)
        return cyc, pairs


def search(seed, iters):
    rng = random.Random(seed)
    while True:
        lay = Layout(rng)
        if lay.colour():
            break
    assert lay.forward_conflicts() == 0
    def score():
        c, p = lay.backward_cost()
        return c + 0.25 * p
    cur = score()
    best = (cur, list(lay.bit_at), list(lay.swap), list(lay.loc), dict(lay.round))
    t0, t1 = 1.0, 0.05
    for it in range(iters):
        temp = t0 * (t1 / t0) ** (it / iters)
        r = rng.random()
        if r < 0.5:
            n = rng.randrange(128)
            lay.swap[n] ^= 1
            new = score()
            if new <= cur or rng.random() < pow(2.718281828, (cur - new) / temp):
                cur = new
            else:
                lay.swap[n] ^= 1
        else:
            # move a whole layout element and recolour: two bits trade places, or two checks trade lanes
            sb, sl, sr = list(lay.bit_at), list(lay.loc), dict(lay.round)
            if r < 0.85:
                a, b = rng.randrange(128), rng.randrange(128)
                lay.bit_at[a], lay.bit_at[b] = lay.bit_at[b], lay.bit_at[a]
            else:
                a, b = rng.randrange(38), rng.randrange(38)
                lay.loc[a], lay.loc[b] = lay.loc[b], lay.loc[a]
            ok = lay.colour()
            new = score() if ok else None
            if ok and (new <= cur or rng.random() < pow(2.718281828, (cur - new) / temp)):
                cur = new
            else:
                lay.bit_at, lay.loc, lay.round = sb, sl, sr
        if cur < best[0]:
            best = (cur, list(lay.bit_at), list(lay.swap), list(lay.loc), dict(lay.round))
    lay.bit_at, lay.swap, lay.loc, lay.round = best[1], best[2], best[3], best[4]
    return lay


def emit(lay):
    cyc, pairs = lay.backward_cost()
    assert lay.forward_conflicts() == 0
    # round of every (check, slot); the unused round of a degree-10 check is the round of its missing slot 10
    rounds = []
    for c, r in enumerate(ROWS):
        used = [lay.round[(c, j)] for j in range(len(r))]
        assert len(set(used)) == len(used)
        free = [k for k in range(ROUNDS) if k not in used]
        rounds.append(used + free)
    nat = Layout(random.Random(0))
    nat.bit_at, nat.swap, nat.loc = list(range(128)), [0] * 128, list(range(38))
    nat.round = {(c, j): j for c, r in enumerate(ROWS) for j in range(len(r))}
    ncyc, _ = nat.backward_cost()
    nfwd = nat.forward_conflicts()
    out = []
    out.append("// GENERATED by tools/layout/make_layout.py (edge colouring + local search over the bit -> lane, check -> lane, first-edge-order")
    out.append("// and round assignments); tests/test_ldpc_layout.py re-derives the conflict counts from these tables with the LDS bank rules")
    out.append("// of MI355X_MICROARCH.md (ds_read_b32: two 32-lane groups, 32 banks of 4 bytes, N distinct addresses on a bank = N cycles).")
    out.append("//")
    out.append("// LDS layout of the BP messages (ldpc.hip): lane l owns codeword bits kBitOfLane[0][l] and kBitOfLane[1][l]; check c is")
    out.append("// processed by lane kLaneOfCheck[c]; kSwapFirstEdges[n] = 1: instruction 0 takes bit n's second edge and instruction 1 its")
    out.append("// first - free, their messages are only ever added to each other first ((tov0 + tov1) + tov2).")
    out.append("//   forward (bit -> check): instruction i of half h stores to the bit-major cell (3h + i)*64 + lane (add-TID store, conflict-free);")
    out.append("//     check lane L reads its edge of round r = kRoundOfSlot[c][j] (j = position of the bit in the check's row): the rounds are")
    out.append("//     a proper edge colouring of (check, bit lane mod 32) within each 32-lane group of checks, so every gather is conflict-free;")
    out.append("//   backward (check -> bit): the product of round r goes to cell r*kTileRowStride + L; the bit side gathers its six cells.")
    out.append(f"// Forward gather: {lay.forward_conflicts()} extra LDS cycles per iteration (rounds = row order on the natural layout: {nfwd}).")
    out.append(f"// Backward gather: {cyc} extra LDS cycles per iteration (natural layout, bit n in lane n % 64, rounds = row order: {ncyc}).")
    out.append("#pragma once")
    out.append("")
    out.append("#include <cstdint>")
    out.append("")
    out.append("namespace msk144")
    out.append("{")
    out.append("")
    out.append(f"constexpr int kTileRowStride = {S};")
    out.append("constexpr uint8_t kBitOfLane[2][64] = {")
    for h in range(2):
        out.append("    {" + ", ".join(str(x) for x in lay.bit_at[64 * h:64 * h + 64]) + "},")
    out.append("};")
    out.append("constexpr uint8_t kSwapFirstEdges[128] = {" + ", ".join(str(x) for x in lay.swap) + "};")
    out.append("constexpr uint8_t kLaneOfCheck[38] = {" + ", ".join(str(x) for x in lay.loc) + "};")
    out.append("constexpr uint8_t kRoundOfSlot[38][11] = {")
    for c in range(38):
        out.append("    {" + ", ".join(str(x) for x in rounds[c]) + "},")
    out.append("};")
    out.append("")
    out.append("}  // namespace msk144")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    lay = search(seed, iters)
    sys.stderr.write(f"seed {seed}: backward {lay.backward_cost()} forward {lay.forward_conflicts()}\n")
    sys.stdout.write(emit(lay))
