/* orc_solver.c — CPU oracle: the end-game solver (lib/alt/reversi_solver_cython.pyx:40-127, used by
 * agent/player.py:100-103,150-161 at the root and :237-251 inside simulations).
 * TEST INFRASTRUCTURE (see orc.h).
 *
 * The reference runs an explicit-stack DFS with a dict memo.  Its return value is a function of the
 * position and the mode alone:
 *   f(own, enemy):  best = (-1, -100); for a in legal moves, ascending:
 *                     [non-exact only: if best.score > 0: stop]            (:105)
 *                     play a; if the opponent can move: v = -f(opp view)   (:111-113, 93-100)
 *                             elif the mover can move again: v = +f(same)  (:114-116)
 *                             else v = discs(mover) - discs(opponent)      (:118-121)
 *                     if best.score < v: best = (a, v)                     (strict: first maximum)
 * and the memo (keyed (own, enemy, next_player); next_player never changes f) only saves time: the
 * reference clears it when a player switches from non-exact to exact calls (:50-52) and a player
 * never goes back (once its root turn reaches use_solver_turn every later move is solved exactly),
 * so exact and non-exact entries are never mixed.  This file is that recursion with its own memo
 * per mode; tests/test_oracle_solver.py checks it against the reference's compiled Cython solver.
 */
#include <stdlib.h>
#include <string.h>
#include "orc.h"

typedef struct {
    u64 own, enemy;
    int8_t move, score;
    uint8_t used;
} sentry;

struct orc_solver {
    sentry* e[2]; /* [0] non-exact, [1] exact */
    size_t cap[2], count[2];
    long long nodes;
};

static size_t shash(u64 a, u64 b) {
    u64 x = a * 0x9E3779B97F4A7C15ULL ^ (b + 0x7F4A7C159E3779B9ULL) * 0xC2B2AE3D27D4EB4FULL;
    x ^= x >> 31; x *= 0xD6E8FEB86659FD93ULL; x ^= x >> 29;
    return (size_t)x;
}

orc_solver* orc_solver_new(void) {
    orc_solver* s = (orc_solver*)calloc(1, sizeof *s);
    for (int m = 0; m < 2; ++m) { s->cap[m] = 1 << 12; s->e[m] = (sentry*)calloc(s->cap[m], sizeof(sentry)); }
    return s;
}
void orc_solver_free(orc_solver* s) { if (s) { free(s->e[0]); free(s->e[1]); free(s); } }
long long orc_solver_nodes(const orc_solver* s) { return s->nodes; }

static sentry* sfind(orc_solver* s, int m, u64 own, u64 enemy) {
    size_t i = shash(own, enemy) & (s->cap[m] - 1);
    while (s->e[m][i].used) {
        if (s->e[m][i].own == own && s->e[m][i].enemy == enemy) return &s->e[m][i];
        i = (i + 1) & (s->cap[m] - 1);
    }
    return NULL;
}
static void sput(orc_solver* s, int m, u64 own, u64 enemy, int move, int score) {
    if ((s->count[m] + 1) * 2 > s->cap[m]) {
        sentry* old = s->e[m]; size_t oc = s->cap[m];
        s->cap[m] *= 2; s->e[m] = (sentry*)calloc(s->cap[m], sizeof(sentry)); s->count[m] = 0;
        for (size_t i = 0; i < oc; ++i) if (old[i].used) sput(s, m, old[i].own, old[i].enemy, old[i].move, old[i].score);
        free(old);
    }
    size_t i = shash(own, enemy) & (s->cap[m] - 1);
    while (s->e[m][i].used) i = (i + 1) & (s->cap[m] - 1);
    s->e[m][i].own = own; s->e[m][i].enemy = enemy; s->e[m][i].move = (int8_t)move; s->e[m][i].score = (int8_t)score; s->e[m][i].used = 1;
    s->count[m]++;
}

static void solve_rec(orc_solver* s, int exactly, u64 own, u64 enemy, int* move, int* score) {
    sentry* c = sfind(s, exactly, own, enemy);
    if (c) { *move = c->move; *score = c->score; return; }
    s->nodes++;
    int best_move = -1, best_score = -100;
    u64 legal = orc_find_correct_moves(own, enemy);
    for (int a = 0; a < 64; ++a) {
        if (!(legal >> a & 1)) continue;
        if (!exactly && best_score > 0) break;
        u64 flipped = orc_calc_flip(a, own, enemy);
        u64 nown = (own ^ flipped) | ((u64)1 << a), nenemy = enemy ^ flipped;
        int v, cm;
        if (orc_find_correct_moves(nenemy, nown)) { solve_rec(s, exactly, nenemy, nown, &cm, &v); v = -v; }
        else if (orc_find_correct_moves(nown, nenemy)) solve_rec(s, exactly, nown, nenemy, &cm, &v);
        else v = orc_bit_count(nown) - orc_bit_count(nenemy);
        if (best_score < v) { best_move = a; best_score = v; }
    }
    sput(s, exactly, own, enemy, best_move, best_score);
    *move = best_move; *score = best_score;
}

/* ReversiSolver.solve(black, white, next_player, exactly) (:40-61).  Returns 1 and (*move, *score from the
 * side to move's view) or 0 for the reference's (None, None). */
int orc_solver_solve(orc_solver* s, u64 black, u64 white, int next_player, int exactly, int* move, int* score) {
    u64 own = next_player == 1 ? black : white, enemy = next_player == 1 ? white : black;
    solve_rec(s, exactly ? 1 : 0, own, enemy, move, score);
    return *move >= 0;
}
