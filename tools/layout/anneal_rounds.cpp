// Offline search for the LDPC message-tile layout of ldpc.hip (developer tool, not part of the product; its result goes through
// make_layout.py --from into msk144cudecoder_amd/csrc/ldpc_layout.h, and tests/test_ldpc_layout.py re-derives every count).
//
// One BP iteration moves the 384 edge messages through two per-wave LDS tiles:
//   forward  (bit -> check): instruction i of half h stores to cell (3h + i)*64 + lane (add-TID store, never conflicted); check lane L
//            gathers one factor per round r.  Bank of a real factor = lane of its bit mod 32.  A degree-10 check has one empty
//            round in which it reads a constant 1.0: the tile keeps 32 such cells, one per bank, and every empty read of an
//            (round, 32-lane group) access goes to ONE bank no real read of that access uses (same address = broadcast) - free
//            whenever the real reads are conflict-free, because then fewer than 32 banks are taken.
//   backward (check -> bit): the product of round r goes to cell kRowBase[r] + L (add-TID store); the bit lanes gather the cells
//            of their six edges.  Bank = (kRowBase[r] + L) mod 32.
// Cost = extra LDS cycles of the 11 x 2 forward and 6 x 2 backward gathers (two 32-lane groups per instruction, 32 banks of
// 4 bytes, N distinct addresses on one bank = N cycles) + a smooth term (colliding pairs).
// Freedom: bit -> (half, lane); which of a bit's first two edges instruction 0 takes; check -> lane; the round of every edge at
// its check (a permutation per check); the residue of every backward row base.
//   g++ -O2 -o anneal_rounds anneal_rounds.cpp && ./anneal_rounds edges.txt seed iterations > layout.txt
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

static int EJ[128][3], EC[128][3];
static int DEG[38];

struct State
{
    int bit_at[128];  // position h*64 + lane -> bit
    int swp[128];     // per bit
    int loc[38];      // check -> lane
    int rnd[38][11];  // check, slot j -> round (slot 10 of a degree-10 check = its empty round)
    int rowb[11];     // residue mod 32 of the backward row base of round r
};

struct Cost
{
    int fwd, bwd, pf, pb;
    double total(double wf) const { return wf * (fwd + 0.25 * pf) + (bwd + 0.25 * pb); }
};

static Cost cost(const State& s)
{
    Cost k{0, 0, 0, 0};
    int lane_of_bit[128];
    for(int p = 0; p < 128; p++) lane_of_bit[s.bit_at[p]] = p & 63;
    // forward: per (round, group of check lanes) the banks of the real factors
    int cf[11][2][32];
    memset(cf, 0, sizeof(cf));
    for(int n = 0; n < 128; n++)
        for(int e = 0; e < 3; e++)
        {
            const int c = EC[n][e], j = EJ[n][e];
            cf[s.rnd[c][j]][s.loc[c] >= 32][lane_of_bit[n] & 31]++;
        }
    for(int r = 0; r < 11; r++)
        for(int g = 0; g < 2; g++)
        {
            int m = 1;
            for(int b = 0; b < 32; b++)
            {
                m = std::max(m, cf[r][g][b]);
                k.pf += cf[r][g][b] * (cf[r][g][b] - 1) / 2;
            }
            k.fwd += m - 1;
        }
    for(int h = 0; h < 2; h++)
        for(int i = 0; i < 3; i++)
            for(int g = 0; g < 2; g++)
            {
                int cb[32] = {0};
                int m = 1;
                for(int l = 32 * g; l < 32 * g + 32; l++)
                {
                    const int n = s.bit_at[h * 64 + l];
                    const int e = (i < 2 && s.swp[n]) ? 1 - i : i;
                    const int c = EC[n][e];
                    const int b = (s.rowb[s.rnd[c][EJ[n][e]]] + s.loc[c]) & 31;
                    cb[b]++;
                    m = std::max(m, cb[b]);
                }
                k.bwd += m - 1;
                for(int b = 0; b < 32; b++) k.pb += cb[b] * (cb[b] - 1) / 2;
            }
    return k;
}

int main(int argc, char** argv)
{
    if(argc < 4) return 2;
    FILE* f = fopen(argv[1], "r");
    if(!f) return 1;
    for(int n = 0; n < 128; n++)
        for(int e = 0; e < 3; e++)
        {
            if(fscanf(f, "%d %d", &EJ[n][e], &EC[n][e]) != 2) return 1;
            DEG[EC[n][e]] = std::max(DEG[EC[n][e]], EJ[n][e] + 1);
        }
    const unsigned seed = atoi(argv[2]);
    const long iters = atol(argv[3]);
    const double wf = argc > 4 ? atof(argv[4]) : 2.0;
    std::mt19937 rng(seed);
    State s;
    for(int i = 0; i < 128; i++) s.bit_at[i] = i;
    std::shuffle(s.bit_at, s.bit_at + 128, rng);
    for(int i = 0; i < 128; i++) s.swp[i] = rng() & 1;
    for(int i = 0; i < 38; i++) s.loc[i] = i;
    std::shuffle(s.loc, s.loc + 38, rng);
    for(int c = 0; c < 38; c++)
    {
        for(int j = 0; j < 11; j++) s.rnd[c][j] = j;
        std::shuffle(s.rnd[c], s.rnd[c] + 11, rng);
    }
    for(int r = 0; r < 11; r++) s.rowb[r] = rng() & 31;
    double cur = cost(s).total(wf);
    State best = s;
    double best_cost = cur;
    std::uniform_real_distribution<double> U(0.0, 1.0);
    const double T0 = 1.2, T1 = 0.04;
    for(long it = 0; it < iters && best_cost > 0; it++)
    {
        const double T = T0 * std::pow(T1 / T0, double(it) / iters);
        State t = s;
        const double r = U(rng);
        if(r < 0.35)
            std::swap(t.bit_at[rng() % 128], t.bit_at[rng() % 128]);
        else if(r < 0.5)
            t.swp[rng() % 128] ^= 1;
        else if(r < 0.6)
            std::swap(t.loc[rng() % 38], t.loc[rng() % 38]);
        else if(r < 0.97)
        {
            const int c = rng() % 38;
            std::swap(t.rnd[c][rng() % 11], t.rnd[c][rng() % 11]);
        }
        else
            t.rowb[rng() % 11] = rng() & 31;
        const double c = cost(t).total(wf);
        if(c <= cur || U(rng) < std::exp((cur - c) / T))
        {
            s = t;
            cur = c;
            if(cur < best_cost)
            {
                best_cost = cur;
                best = s;
            }
        }
    }
    const Cost k = cost(best);
    printf("seed %u forward %d backward %d\n", seed, k.fwd, k.bwd);
    printf("bits");
    for(int i = 0; i < 128; i++) printf(" %d", best.bit_at[i]);
    printf("\nswap");
    for(int i = 0; i < 128; i++) printf(" %d", best.swp[i]);
    printf("\nloc");
    for(int i = 0; i < 38; i++) printf(" %d", best.loc[i]);
    printf("\nrounds");
    for(int c = 0; c < 38; c++)
        for(int j = 0; j < 11; j++) printf(" %d", best.rnd[c][j]);
    printf("\nrowres");
    for(int r = 0; r < 11; r++) printf(" %d", best.rowb[r]);
    printf("\n");
    return 0;
}
