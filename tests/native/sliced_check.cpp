// tests/native/sliced_check.cpp - raz_sweep_sliced.h (the bit-sliced find_correct_moves / calc_flip / step of the large-batch sweep
// kernels: 32 boards per lane, one bit per board) against raz_bitboard.h (the statement of lib/bitboard.py / env/reversi_env.py that
// the oracle and the goldens pin), on the host: the header's plain-C++ branches, one "lane" at a time - positions from random playouts,
// random garbage (overlapping colours, full and sparse boards), every kind of action.
// Built and run by tests/test_native_host.py::test_bit_sliced_sweep_arithmetic_equals_the_reference_shaped_ops.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define __device__
#define __forceinline__ inline
struct ulonglong2 { unsigned long long x, y; };   // (the load / store helpers of the header: not used here)
#include "../../reversi-alpha-zero_amd/csrc/raz_sweep_sliced.h"

static unsigned long long seed_ = 0x9E3779B97F4A7C15ULL;
static unsigned long long rnd() {   // splitmix64
    unsigned long long z = (seed_ += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

struct Boards { raz_bb black[32], white[32]; unsigned char player[32], status[32], action[32]; };

static void to_words(const raz_bb* x, uint32_t (&a)[64]) {
    for (int k = 0; k < 32; ++k) { a[k] = (uint32_t)x[k]; a[32 + k] = (uint32_t)(x[k] >> 32); }
}
static void from_words(const uint32_t (&a)[64], raz_bb* x) {
    for (int k = 0; k < 32; ++k) x[k] = ((raz_bb)a[32 + k] << 32) | a[k];
}

static long long checked = 0;
static int check_lane(const Boards& in, int what) {
    uint32_t b[64], w[64], L[64];
    to_words(in.black, b);
    to_words(in.white, w);
    // transposes are involutions and really transpose
    {
        uint32_t t[64];
        memcpy(t, b, sizeof t);
        sl::transpose_boards(t);
        for (int s = 0; s < 64; ++s)
            for (int k = 0; k < 32; ++k)
                if (((t[s] >> k) & 1u) != ((in.black[k] >> s) & 1ULL)) { printf("transpose: square %d board %d\n", s, k); return 1; }
        sl::transpose_boards(t);
        if (memcmp(t, b, sizeof t)) { printf("transpose is not an involution\n"); return 1; }
    }
    if (what == 0) {   // find_correct_moves: black as own, white as enemy
        sl::transpose_boards(b);
        sl::transpose_boards(w);
        sl::mobility(b, w, L);
        sl::transpose_boards(L);
        raz_bb got[32];
        from_words(L, got);
        for (int k = 0; k < 32; ++k) {
            ++checked;
            const raz_bb want = bb_legal_moves(in.black[k], in.white[k]);
            if (got[k] != want) { printf("legal_moves differ: own %016llx enemy %016llx: %016llx vs %016llx\n", in.black[k], in.white[k], want, got[k]); return 1; }
        }
        return 0;
    }
    uint32_t pw[8], sw[8], aw[8];
    for (int q = 0; q < 8; ++q) {
        pw[q] = sw[q] = aw[q] = 0;
        for (int i = 0; i < 4; ++i) {
            pw[q] |= (uint32_t)in.player[4 * q + i] << (8 * i);
            sw[q] |= (uint32_t)in.status[4 * q + i] << (8 * i);
            aw[q] |= (uint32_t)in.action[4 * q + i] << (8 * i);
        }
    }
    sl::transpose_boards(b);
    sl::transpose_boards(w);
    const sl::StepMasks m = sl::step_boards(b, w, L, pw, sw, aw);
    sl::transpose_boards(b);
    sl::transpose_boards(w);
    sl::transpose_boards(L);
    raz_bb nb[32], nw[32], nl[32];
    from_words(b, nb);
    from_words(w, nw);
    from_words(L, nl);
    for (int k = 0; k < 32; ++k) {
        ++checked;
        const bool overlapping = (in.black[k] & in.white[k]) != 0;
        if (overlapping != (((m.overlap >> k) & 1u) != 0)) { printf("overlap flag of board %d: %d\n", k, (int)((m.overlap >> k) & 1u)); return 1; }
        if (overlapping) continue;   // (not a position: the kernel steps such a lane board by board - raz_sweep_sliced.h StepMasks)
        raz_step_result want;
        if (in.status[k] != 0) {   // the sweep's contract: a finished game is left alone, legal = 0
            want.black = in.black[k]; want.white = in.white[k]; want.player = in.player[k]; want.status = in.status[k]; want.legal = 0;
        } else if (in.action[k] >= 64 && in.action[k] != RAZ_ACTION_RESIGN) {   // outside the board: a move that flips nothing
            want.black = in.black[k]; want.white = in.white[k]; want.player = in.player[k]; want.legal = 0;
            want.status = (unsigned char)((in.player[k] == RAZ_PLAYER_BLACK ? RAZ_WIN_WHITE : RAZ_WIN_BLACK) | RAZ_STATUS_ILLEGAL);
        } else
            want = bb_env_step(in.black[k], in.white[k], in.player[k], in.action[k]);
        // what the kernel's epilogue derives from the masks
        unsigned p2 = in.player[k], s2 = in.status[k];
        const unsigned other_wins = in.player[k] == RAZ_PLAYER_BLACK ? RAZ_WIN_WHITE : RAZ_WIN_BLACK;
        const bool moved = (m.moved >> k) & 1u, nz1 = (m.nz1 >> k) & 1u, nz2 = (m.nz2 >> k) & 1u;
        if (in.status[k] == 0) {
            if (in.action[k] == RAZ_ACTION_RESIGN) s2 = other_wins | RAZ_STATUS_RESIGNED;
            else if (!moved) s2 = other_wins | RAZ_STATUS_ILLEGAL;
            else if (nz1) p2 = (3u - p2) & 0xffu;
            else if (!nz2) {
                const int cb = bb_popcount(nb[k]), cw = bb_popcount(nw[k]);
                s2 = cb > cw ? RAZ_WIN_BLACK : (cb < cw ? RAZ_WIN_WHITE : RAZ_WIN_DRAW);
            }
        }
        if (nb[k] != want.black || nw[k] != want.white || nl[k] != want.legal || p2 != want.player || s2 != want.status) {
            printf("step differs: black %016llx white %016llx player %d status %d action %d\n  want black %016llx white %016llx legal %016llx player %d status %d\n"
                   "  got  black %016llx white %016llx legal %016llx player %u status %u (moved %d nz1 %d nz2 %d)\n",
                   in.black[k], in.white[k], in.player[k], in.status[k], in.action[k], want.black, want.white, want.legal, want.player, want.status,
                   nb[k], nw[k], nl[k], p2, s2, (int)moved, (int)nz1, (int)nz2);
            return 1;
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    const long long lanes = argc > 1 ? atoll(argv[1]) : 20000;
    Boards B;
    for (long long it = 0; it < lanes; ++it) {
        const int kind = (int)(it % 6);
        for (int k = 0; k < 32; ++k) {
            raz_bb b, w;
            int player = 1 + (int)(rnd() & 1);
            if (kind <= 1) {   // a playout position
                b = 0x0000000810000000ULL; w = 0x0000001008000000ULL; player = 1;
                int plies = (int)(rnd() % 61), st = 0;
                for (int i = 0; i < plies && !st; ++i) {
                    const raz_bb own = player == 1 ? b : w, en = player == 1 ? w : b;
                    raz_bb lg = bb_legal_moves(own, en);
                    int n = bb_popcount(lg), pick = (int)(rnd() % (unsigned)(n ? n : 1));
                    if (!n) break;
                    for (int j = 0; j < pick; ++j) lg &= lg - 1;
                    const raz_step_result r = bb_env_step(b, w, player, __builtin_ctzll(lg));
                    if (r.status) break;
                    b = r.black; w = r.white; player = r.player;
                }
            } else if (kind == 2) { b = rnd(); w = rnd() & ~b; }
            else if (kind == 3) { b = rnd(); w = rnd(); }                                   // overlapping garbage
            else if (kind == 4) { b = rnd() & rnd() & rnd(); w = rnd() & rnd() & ~b; }      // sparse
            else { b = rnd() | rnd(); w = ~b | (rnd() & rnd()); }                           // full, overlapping
            B.black[k] = b; B.white[k] = w; B.player[k] = (unsigned char)player;
            B.status[k] = (rnd() % 10 == 0) ? (unsigned char)(1 + rnd() % 3) : 0;
            const raz_bb own = player == 1 ? b : w, en = player == 1 ? w : b;
            raz_bb lg = bb_legal_moves(own, en);
            const unsigned r = (unsigned)(rnd() % 16);
            if (lg && r < 11) {   // a legal move
                int pick = (int)(rnd() % (unsigned)bb_popcount(lg));
                for (int j = 0; j < pick; ++j) lg &= lg - 1;
                B.action[k] = (unsigned char)__builtin_ctzll(lg);
            } else if (r == 11) B.action[k] = RAZ_ACTION_RESIGN;
            else if (r == 12) B.action[k] = (unsigned char)(64 + rnd() % 190);
            else B.action[k] = (unsigned char)(rnd() % 64);
        }
        if (check_lane(B, 0) || check_lane(B, 1)) return 1;
    }
    printf("SLICED_OK %lld boards\n", checked);
    return 0;
}
