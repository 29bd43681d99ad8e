// tests/native/bbv_check.cpp - raz_bitboard_valu.h (the VALU-shaped mobility / step used by the sweep kernels) against
// raz_bitboard.h (the statement of lib/bitboard.py / env/reversi_env.py that the oracle and the goldens pin), on the host:
// positions from random playouts, random garbage (overlapping colours, full boards, sparse boards), every action.
// Built and run by tests/test_native_host.py::test_valu_shaped_bitboard_ops_equal_the_reference_shaped_ones.
#include <cstdio>
#include <cstdlib>
#include "../../reversi-alpha-zero_amd/csrc/raz_bitboard_valu.h"

static unsigned long long s = 0x9E3779B97F4A7C15ULL;
static unsigned long long rnd() {   // splitmix64
    unsigned long long z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static long long checked = 0;
static int check_pair(raz_bb own, raz_bb enemy) {
    const raz_bb a = bb_legal_moves(own, enemy), b = bbv_legal_moves(own, enemy);
    ++checked;
    if (a != b) {
        printf("legal_moves differ: own %016llx enemy %016llx: %016llx vs %016llx\n", own, enemy, a, b);
        return 1;
    }
    return 0;
}
static int check_flip(int pos, raz_bb own, raz_bb enemy) {
    const raz_bb a = bb_calc_flip(pos, own, enemy), b = bbv_calc_flip(pos, own, enemy);
    ++checked;
    if (a != b) {
        printf("calc_flip differ: pos %d own %016llx enemy %016llx: %016llx vs %016llx\n", pos, own, enemy, a, b);
        return 1;
    }
    return 0;
}
static int check_step(raz_bb black, raz_bb white, int player, int action) {
    const raz_step_result a = bb_env_step(black, white, player, action);
    raz_step_result b = bbv_env_step_first(black, white, player, action);
    if (b.status == RAZ_STEP_OPP_STUCK) bbv_env_step_finish(b);
    ++checked;
    if (a.black != b.black || a.white != b.white || a.player != b.player || a.status != b.status || a.legal != b.legal) {
        printf("env_step differ: black %016llx white %016llx player %d action %d: status %d vs %d, legal %016llx vs %016llx\n",
               black, white, player, action, a.status, b.status, a.legal, b.legal);
        return 1;
    }
    return 0;
}

int main(int argc, char** argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 2000000;
    int bad = 0;
    // the reference's own test boards (test/lib/test_bitboard.py) and the start position
    const raz_bb fixed[][2] = {{RAZ_INIT_BLACK, RAZ_INIT_WHITE}, {0x00000000081d0603ULL, 0x0002043814020100ULL},
                               {0x0088ffabd5dfdf5fULL, 0x000700542a202020ULL}, {0xfe71797106000203ULL, 0x008e868ef9fffd7cULL},
                               {0, 0}, {~0ULL, 0}, {0, ~0ULL}, {~0ULL, ~0ULL}};
    for (auto& f : fixed) {
        bad += check_pair(f[0], f[1]) + check_pair(f[1], f[0]);
        for (int a = 0; a < 64; ++a) bad += check_step(f[0], f[1], 1, a) + check_step(f[0], f[1], 2, a) + check_flip(a, f[0], f[1]);
    }
    // random playouts: every position of the game, every action on it (legal or not), resignation
    for (long long g = 0; g < n / 4000 + 1 && !bad; ++g) {
        raz_bb black = RAZ_INIT_BLACK, white = RAZ_INIT_WHITE;
        int player = 1;
        for (int ply = 0; ply < 70; ++ply) {
            const raz_bb own = player == 1 ? black : white, enemy = player == 1 ? white : black;
            bad += check_pair(own, enemy) + check_pair(enemy, own);
            for (int a = 0; a < 64; ++a) bad += check_step(black, white, player, a) + check_flip(a, own, enemy);
            bad += check_step(black, white, player, RAZ_ACTION_RESIGN);
            const raz_bb legal = bb_legal_moves(own, enemy);
            if (!legal) break;
            int k = (int)(rnd() % (unsigned)bb_popcount(legal)), a = 0;
            for (raz_bb m = legal;; m &= m - 1, --k)
                if (k == 0) { a = __builtin_ctzll(m); break; }
            const raz_step_result r = bb_env_step(black, white, player, a);
            black = r.black; white = r.white; player = r.player;
            if (r.status) break;
        }
    }
    // garbage: dense, sparse, overlapping, complementary
    for (long long i = 0; i < n && !bad; ++i) {
        raz_bb a = rnd(), b = rnd();
        switch (i & 7) {
            case 0: break;                                   // dense, overlapping
            case 1: b &= ~a; break;                          // dense, disjoint
            case 2: a &= rnd(); b &= rnd() & ~a; break;      // half density
            case 3: a &= rnd() & rnd(); b &= rnd() & rnd(); break;
            case 4: b = ~a; break;                           // full board
            case 5: a &= rnd() & rnd() & rnd(); b |= rnd(); break;
            case 6: a |= rnd(); b &= rnd() & rnd() & rnd(); break;
            case 7: a = 1ULL << (rnd() & 63); b = rnd() | rnd(); break;
        }
        bad += check_pair(a, b);
        bad += check_flip((int)(rnd() & 63), a, b);
        bad += check_step(a, b, 1 + (int)(i & 1), (int)(rnd() & 63));
    }
    if (bad) return 1;
    printf("BBV_OK %lld\n", checked);
    return 0;
}
