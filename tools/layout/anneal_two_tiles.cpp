// Offline search for the LDPC message-tile layout of ldpc.hip (not part of the product; its result is ldpc_layout.h).
// Two tiles: the FORWARD tile (bit -> check) holds slot pairs (j, j+1) of a check side by side so that a check lane reads its
// column with six ds_read_b64: cell = (j/2)*S1 + 2*lane_of_check + (j&1); the BACKWARD tile (check -> bit) keeps
// cell = j*S2 + lane_of_check for the M0-relative column stores.  Cost = extra LDS-array cycles of the twelve edge-side
// instructions (six scatter stores into the forward tile, six gather loads from the backward tile), each serviced as two
// groups of 32 lanes, 32 banks of 4 bytes, N distinct addresses on one bank = N cycles.
// Freedom: which two codeword bits a lane owns (128 positions), which of a bit's first two edges instruction 0 takes
// ((tov0 + tov1) + tov2 is commutative in them), which lane walks which check's column.
//   g++ -O2 -o anneal anneal_two_tiles.cpp && ./anneal edges.txt S1 S2 seed iterations
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

static int EJ[128][3], EC[128][3];
static int S1, S2;
static int pairs_f, pairs_b;  // colliding lane pairs of the last cost() call: the smooth part of the objective

struct State
{
    int bit_at[128];  // position h*64 + lane -> bit
    int swp[128];     // per bit
    int loc[38];      // check -> lane
};

static void cost(const State& s, int& fwd, int& bwd, int& worst)
{
    fwd = bwd = 0;
    pairs_f = pairs_b = 0;
    worst = 1;
    for(int h = 0; h < 2; h++)
        for(int i = 0; i < 3; i++)
            for(int g = 0; g < 2; g++)
            {
                int cf[32] = {0}, cb[32] = {0};
                int af[32][4], ab[32][4];
                int mf = 1, mb = 1;
                for(int l = 32 * g; l < 32 * g + 32; l++)
                {
                    const int n = s.bit_at[h * 64 + l];
                    const int k = (i < 2 && s.swp[n]) ? 1 - i : i;
                    const int j = EJ[n][k], c = EC[n][k];
                    const int a1 = (j >> 1) * S1 + 2 * s.loc[c] + (j & 1);
                    const int a2 = j * S2 + s.loc[c];
                    int b = a1 & 31;
                    af[b][cf[b] & 3] = a1;
                    cf[b]++;
                    if(cf[b] > mf) mf = cf[b];
                    b = a2 & 31;
                    ab[b][cb[b] & 3] = a2;
                    cb[b]++;
                    if(cb[b] > mb) mb = cb[b];
                }
                (void)af;
                (void)ab;  // all cells of one instruction are distinct edges: distinct addresses
                fwd += mf - 1;
                bwd += mb - 1;
                for(int q = 0; q < 32; q++)
                {
                    pairs_f += cf[q] * (cf[q] - 1) / 2;
                    pairs_b += cb[q] * (cb[q] - 1) / 2;
                }
                worst = std::max(worst, std::max(mf, mb));
            }
}

int main(int argc, char** argv)
{
    FILE* f = fopen(argv[1], "r");
    for(int n = 0; n < 128; n++)
        for(int k = 0; k < 3; k++)
            if(fscanf(f, "%d %d", &EJ[n][k], &EC[n][k]) != 2) return 1;
    S1 = atoi(argv[2]);
    S2 = atoi(argv[3]);
    const unsigned seed = atoi(argv[4]);
    const long iters = atol(argv[5]);
    const double wf = argc > 6 ? atof(argv[6]) : 1.0;  // weight of forward (store) conflicts
    std::mt19937 rng(seed);
    State s;
    for(int i = 0; i < 128; i++) s.bit_at[i] = i;
    std::shuffle(s.bit_at, s.bit_at + 128, rng);
    for(int i = 0; i < 128; i++) s.swp[i] = rng() & 1;
    for(int i = 0; i < 38; i++) s.loc[i] = i;
    std::shuffle(s.loc, s.loc + 38, rng);
    int cf, cb, w;
    cost(s, cf, cb, w);
    double cur = wf * (cf + 0.25 * pairs_f) + (cb + 0.25 * pairs_b);
    State best = s;
    double best_cost = cur;
    std::uniform_real_distribution<double> U(0.0, 1.0);
    const double T0 = 1.5, T1 = 0.08;
    for(long it = 0; it < iters; it++)
    {
        const double T = T0 * std::pow(T1 / T0, double(it) / iters);
        State t = s;
        const double r = U(rng);
        if(r < 0.6)
            std::swap(t.bit_at[rng() % 128], t.bit_at[rng() % 128]);
        else if(r < 0.85)
            t.swp[rng() % 128] ^= 1;
        else
            std::swap(t.loc[rng() % 38], t.loc[rng() % 38]);
        cost(t, cf, cb, w);
        const double c = wf * (cf + 0.25 * pairs_f) + (cb + 0.25 * pairs_b);
        if(c <= cur || U(rng) < std::exp((cur - c) / T))
        {
            s = t;
            cur = c;
            if(cur < best_cost)
            {
                best_cost = cur;
                best = s;
                if(best_cost == 0) break;
            }
        }
    }
    cost(best, cf, cb, w);
    printf("S1 %d S2 %d seed %u forward %d backward %d worst %d\n", S1, S2, seed, cf, cb, w);
    printf("bits");
    for(int i = 0; i < 128; i++) printf(" %d", best.bit_at[i]);
    printf("\nswap");
    for(int i = 0; i < 128; i++) printf(" %d", best.swp[i]);
    printf("\nloc");
    for(int i = 0; i < 38; i++) printf(" %d", best.loc[i]);
    printf("\n");
    return 0;
}
