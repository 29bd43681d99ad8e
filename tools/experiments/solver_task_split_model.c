// How busy would 64 lanes be?  Tasks = the subtrees 2 (or 3) plies below the root of a non-exact solve, handed to idle lanes in the
// reference's scan order; duration = nodes of the faithful non-exact scan of that subtree.  Only the tasks the sequential scan would
// have visited count (the others are speculative and get cancelled); greedy list scheduling in order.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "orc.h"
static long long nodes;
static int pc(u64 x) { return __builtin_popcountll(x); }
#define MAXT 4096
static long long dur[MAXT]; static int nt;
// faithful non-exact scan; at depth == split the subtree's node count is recorded as one task
static int f_ne(u64 own, u64 enemy, int depth, int split) {
    long long n0 = nodes;
    nodes++;
    u64 legal = orc_find_correct_moves(own, enemy);
    int v;
    if (!legal) {
        u64 l2 = orc_find_correct_moves(enemy, own);
        if (!l2) v = pc(own) - pc(enemy);
        else v = -f_ne(enemy, own, depth, split);      // a pass does not count as a ply of the split
    } else {
        int bs = -100;
        for (u64 m = legal; m; m &= m - 1) {
            int a = __builtin_ctzll(m);
            u64 fl = orc_calc_flip(a, own, enemy);
            int c = -f_ne(enemy ^ fl, (own ^ fl) | (1ULL << a), depth + 1, split);
            if (bs < c) bs = c;
            if (bs > 0) break;
        }
        v = bs;
    }
    if (depth == split && nt < MAXT) dur[nt++] = nodes - n0;
    return v;
}
static double makespan(int lanes) {
    long long t[64]; memset(t, 0, sizeof t);
    for (int i = 0; i < nt; ++i) { int b = 0; for (int l = 1; l < lanes; ++l) if (t[l] < t[b]) b = l; t[b] += dur[i]; }
    long long m = 0; for (int l = 0; l < lanes; ++l) if (t[l] > m) m = t[l];
    return (double)m;
}
int main(int argc, char** argv) {
    int empties = argc > 1 ? atoi(argv[1]) : 10, N = argc > 2 ? atoi(argv[2]) : 200;
    srand(777);
    double tot = 0, ms2 = 0, ms3 = 0, ms4 = 0; int done = 0; double t2 = 0, t3 = 0, t4 = 0;
    while (done < N) {
        orc_env e; orc_env_reset(&e);
        while (!e.done && 64 - pc(e.black | e.white) > empties) {
            u64 own = e.next_player == 1 ? e.black : e.white, en = e.next_player == 1 ? e.white : e.black;
            u64 legal = orc_find_correct_moves(own, en);
            int k = pc(legal), r = rand() % k; u64 m = legal; while (r--) m &= m - 1;
            orc_env_step(&e, __builtin_ctzll(m));
        }
        if (e.done || 64 - pc(e.black | e.white) != empties) continue;
        u64 own = e.next_player == 1 ? e.black : e.white, en = e.next_player == 1 ? e.white : e.black;
        for (int split = 2; split <= 4; ++split) {
            nodes = 0; nt = 0; f_ne(own, en, 0, split);
            double m = makespan(64);
            if (split == 2) { tot += nodes; ms2 += m; t2 += nt; } else if (split == 3) { ms3 += m; t3 += nt; } else { ms4 += m; t4 += nt; }
        }
        done++;
    }
    printf("empties %d, %d positions: nodes/solve %.0f\n", empties, N, tot / N);
    printf("2-ply tasks: %.1f needed tasks/solve, makespan %.0f nodes, lane utilisation %.1f %%\n", t2 / N, ms2 / N, 100.0 * tot / (64.0 * ms2));
    printf("3-ply tasks: %.1f needed tasks/solve, makespan %.0f nodes, lane utilisation %.1f %%, speed-up over 2-ply %.2f\n", t3 / N, ms3 / N, 100.0 * tot / (64.0 * ms3), ms2 / ms3);
    printf("4-ply tasks: %.1f needed tasks/solve, makespan %.0f nodes, lane utilisation %.1f %%, speed-up over 2-ply %.2f\n", t4 / N, ms4 / N, 100.0 * tot / (64.0 * ms4), ms2 / ms4);
    return 0;
}
