// Node-count experiment: the reference's non-exact scan (no memo) against a win/loss/draw search that answers only
// "first winning move in ascending order" - how much of the solver's work would it remove?
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "orc.h"
static long long nodes;
static int pc(u64 x) { return __builtin_popcountll(x); }
// faithful non-exact value of (own, enemy), mover = own
static int f_ne(u64 own, u64 enemy, int* move) {
    nodes++;
    u64 legal = orc_find_correct_moves(own, enemy);
    if (!legal) {
        u64 l2 = orc_find_correct_moves(enemy, own);
        if (!l2) { *move = -1; return pc(own) - pc(enemy); }
        int m; int v = -f_ne(enemy, own, &m); *move = -1; return v;   // pass (the reference handles passes inside the child loop; count the same nodes)
    }
    int bm = -1, bs = -100;
    for (u64 m = legal; m; m &= m - 1) {
        int a = __builtin_ctzll(m);
        u64 fl = orc_calc_flip(a, own, enemy);
        int mm; int v = -f_ne(enemy ^ fl, (own ^ fl) | (1ULL << a), &mm);
        if (bs < v) { bs = v; bm = a; }
        if (bs > 0) break;
    }
    *move = bm; return bs;
}
// boolean: can the mover reach value >= need (need = 1: win, need = 0: at least a draw)?
static int can(u64 own, u64 enemy, int need) {
    nodes++;
    u64 legal = orc_find_correct_moves(own, enemy);
    if (!legal) {
        u64 l2 = orc_find_correct_moves(enemy, own);
        if (!l2) return pc(own) - pc(enemy) >= need;
        return !can(enemy, own, 1 - need);   // my value >= need  <=>  opponent's value <= -need  <=>  NOT (opponent >= 1 - need)
    }
    for (u64 m = legal; m; m &= m - 1) {
        int a = __builtin_ctzll(m);
        u64 fl = orc_calc_flip(a, own, enemy);
        if (!can(enemy ^ fl, (own ^ fl) | (1ULL << a), 1 - need)) return 1;
    }
    return 0;
}
int main(int argc, char** argv) {
    int empties = argc > 1 ? atoi(argv[1]) : 10, N = argc > 2 ? atoi(argv[2]) : 200;
    srand(12345);
    long long n_f_win = 0, n_f_lose = 0, n_w_win = 0, n_w_lose = 0; int wins = 0, total = 0, mismatch = 0;
    while (total < N) {
        orc_env e; orc_env_reset(&e);
        while (!e.done && 64 - pc(e.black | e.white) > empties) {
            u64 own = e.next_player == 1 ? e.black : e.white, en = e.next_player == 1 ? e.white : e.black;
            u64 legal = orc_find_correct_moves(own, en);
            int k = pc(legal), r = rand() % k; u64 m = legal; while (r--) m &= m - 1;
            orc_env_step(&e, __builtin_ctzll(m));
        }
        if (e.done || 64 - pc(e.black | e.white) != empties) continue;
        u64 own = e.next_player == 1 ? e.black : e.white, en = e.next_player == 1 ? e.white : e.black;
        int mv; nodes = 0; int v = f_ne(own, en, &mv); long long nf = nodes;
        // WLD: first winning move ascending
        nodes = 0; int wm = -1;
        u64 legal = orc_find_correct_moves(own, en);
        for (u64 m = legal; m; m &= m - 1) {
            int a = __builtin_ctzll(m);
            u64 fl = orc_calc_flip(a, own, en);
            if (!can(en ^ fl, (own ^ fl) | (1ULL << a), 0)) { wm = a; break; }
        }
        long long nw = nodes;
        if ((v > 0) != (wm >= 0) || (v > 0 && wm != mv)) mismatch++;
        if (v > 0) { wins++; n_f_win += nf; n_w_win += nw; } else { n_f_lose += nf; n_w_lose += nw; }
        total++;
    }
    printf("empties %d positions %d wins %d mismatches %d\n", empties, total, wins, mismatch);
    printf("faithful nodes: winning roots %lld, other roots %lld\n", n_f_win, n_f_lose);
    printf("WLD nodes:      winning roots %lld, other roots %lld\n", n_w_win, n_w_lose);
    printf("today %lld -> with WLD first %lld (x%.2f less)\n", n_f_win + n_f_lose, n_w_win + n_w_lose + n_f_lose,
           (double)(n_f_win + n_f_lose) / (double)(n_w_win + n_w_lose + n_f_lose));
    return 0;
}
