"""Offline search for the LDS layout of the BP decoder's message tile (csrc/ldpc_layout.h).

The decoder keeps one float2 cell (codewords A and B of the wave) per Tanner-graph edge: cell(j, c) = j*S + lane_of_check[c]
for slot j of check c.  Check lanes (lane_of_check[c] < 38) walk their column with consecutive cells per row - conflict-free by
construction for ds_read_b64 / ds_write_b64.  The EDGE side is a scatter/gather fixed by the graph: lane l owns codeword bits
bit_of_lane[0][l], bit_of_lane[1][l] and touches the cells of their 3 + 3 edges with six ds_write_b64 + six ds_read_b64 per
iteration.  LDS rules (MI355X_MICROARCH.md, LDS table):
    ds_read_b64   two 32-lane groups, 64 banks of 4 B: no conflict iff the 32 cell indices are distinct mod 32
    ds_write_b64  four 16-lane groups, 32 banks:        no conflict iff the 16 cell indices are distinct mod 16
Free parameters: the row stride S, the bit -> (lane, half) assignment and the check -> lane assignment.  This script anneals
them to zero read conflicts and as few write conflicts as possible and prints the header body.
"""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msk144cudecoder_amd import protocol as P  # noqa: E402

ROWS = [[n for n in r if n >= 0] for r in P.CHECK_BITS]
EDGES = [[] for _ in range(128)]            # bit -> [(slot, check)] ascending check = edge order k
for c, r in enumerate(ROWS):
    for j, n in enumerate(r):
        EDGES[n].append((j, c))


def cost(bit_of, lane_of_check, S, detail=False):
    """bit_of: int[2][64]; returns (read_conflicts, write_conflicts) = extra LDS cycles per iteration."""
    rd = wr = 0
    for h in range(2):
        for k in range(3):
            cells = np.array([EDGES[bit_of[h][l]][k][0] * S + lane_of_check[EDGES[bit_of[h][l]][k][1]] for l in range(64)])
            for g in range(2):
                m = cells[32 * g:32 * g + 32] % 32
                rd += int(np.bincount(m, minlength=32).max() - 1)
            for g in range(4):
                m = cells[16 * g:16 * g + 16] % 16
                wr += int(np.bincount(m, minlength=16).max() - 1)
    return rd, wr


def anneal(S, seed, iters=200000):
    rng = np.random.default_rng(seed)
    perm = rng.permutation(128)
    bit_of = [list(perm[:64]), list(perm[64:])]
    loc = list(rng.permutation(38))          # lane_of_check
    def score(b, lc):
        r, w = cost(b, lc, S)
        return 8 * r + w
    cur = score(bit_of, loc)
    T = 2.0
    for it in range(iters):
        if cur == 0:
            break
        T = max(0.05, T * 0.99997)
        if rng.random() < 0.85:
            a, b = rng.integers(0, 128, 2)
            ha, la, hb, lb = a // 64, a % 64, b // 64, b % 64
            bit_of[ha][la], bit_of[hb][lb] = bit_of[hb][lb], bit_of[ha][la]
            new = score(bit_of, loc)
            if new <= cur or rng.random() < np.exp((cur - new) / T):
                cur = new
            else:
                bit_of[ha][la], bit_of[hb][lb] = bit_of[hb][lb], bit_of[ha][la]
        else:
            a, b = rng.integers(0, 38, 2)
            loc[a], loc[b] = loc[b], loc[a]
            new = score(bit_of, loc)
            if new <= cur or rng.random() < np.exp((cur - new) / T):
                cur = new
            else:
                loc[a], loc[b] = loc[b], loc[a]
    return cur, bit_of, loc


if __name__ == "__main__":
    best = None
    for S in (int(a) for a in sys.argv[1:]) if len(sys.argv) > 1 else range(38, 50):
        for seed in range(2):
            sc, b, lc = anneal(S, seed, iters=int(os.environ.get("ITERS", "60000")))
            r, w = cost(b, lc, S)
            print(f"S={S} seed={seed}: read conflicts {r}, write conflicts {w}", flush=True)
            if best is None or (r, w) < best[0]:
                best = ((r, w), S, b, lc)
    (r, w), S, b, lc = best
    print(f"// read conflict cycles per iteration: {r}, write conflict cycles: {w}")
    print(f"constexpr int kTileRowStride = {S};")
    for h in range(2):
        print(f"// bit_of_lane[{h}]\n    {{" + ", ".join(str(int(x)) for x in b[h]) + "},")
    print("// lane_of_check\n    {" + ", ".join(str(int(x)) for x in lc) + "}")
