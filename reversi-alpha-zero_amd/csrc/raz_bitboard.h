// raz_bitboard.h — 8x8 Reversi bitboard primitives for gfx950 device code and the host-side
// scalar entry points of libraz.  One u64 per colour, bit i = square i, bit 0 = top-left,
// bit 63 = bottom-right (reference convention: lib/bitboard.py:10-17).
//
// Every function here is defined on ALL 2^64 x 2^64 inputs (not only on legal positions) and
// returns exactly what the reference's big-int Python arithmetic returns after its final
// `& own` / `& mask` / `blank &` clamps, so that garbage-in behaviour (occupied square, overlapping
// colours) is the reference's too.  Reference: lib/bitboard.py:53-159 (+ the author's own native
// precedent lib/alt/bitboard_cython.pyx:1-102).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RAZ_HD __host__ __device__ __forceinline__
#else
#define RAZ_HD static inline
#endif

typedef unsigned long long raz_bb;  // same width as uint64_t; matches HIP ulonglong2 lanes

#define RAZ_MASK_LR 0x7e7e7e7e7e7e7e7eULL  // columns 1..6 (lib/bitboard.py:55)
#define RAZ_MASK_TB 0x00ffffffffffff00ULL  // rows 1..6    (lib/bitboard.py:56)
#define RAZ_MASK_IN (RAZ_MASK_LR & RAZ_MASK_TB)

#define RAZ_INIT_BLACK 0x0000000810000000ULL  // env/reversi_env.py:135
#define RAZ_INIT_WHITE 0x0000001008000000ULL  // env/reversi_env.py:136

RAZ_HD int bb_popcount(raz_bb x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}

// 180-degree rotation sends square i to 63-i, i.e. it is a plain 64-bit bit reversal
// (lib/bitboard.py:158-159 builds it from two rotate90s; v_bfrev_b32 x2 on gfx950).
RAZ_HD raz_bb bb_rotate180(raz_bb x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(x);
#else
    x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0f0f0f0f0f0f0f0fULL) | ((x & 0x0f0f0f0f0f0f0f0fULL) << 4);
    return __builtin_bswap64(x);
#endif
}

// Row r <-> row 7-r (== np.flipud on the (8,8) view): a byte swap.  lib/bitboard.py:119-125.
RAZ_HD raz_bb bb_flip_vertical(raz_bb x) { return __builtin_bswap64(x); }

// Transpose about the a1-h8 diagonal (== ndarray.T on the (8,8) view): three delta swaps.
// lib/bitboard.py:141-151.
RAZ_HD raz_bb bb_flip_diag_a1h8(raz_bb x) {
    raz_bb t;
    t = 0x0f0f0f0f00000000ULL & (x ^ (x << 28));
    x ^= t ^ (t >> 28);
    t = 0x3333000033330000ULL & (x ^ (x << 14));
    x ^= t ^ (t >> 14);
    t = 0x5500550055005500ULL & (x ^ (x << 7));
    x ^= t ^ (t >> 7);
    return x;
}

// Rotate the board RIGHT once (== np.rot90(k=-1)).  lib/bitboard.py:154-155.
RAZ_HD raz_bb bb_rotate90(raz_bb x) { return bb_flip_diag_a1h8(bb_flip_vertical(x)); }

// The dihedral transform the player applies before the net and to saved rows:
// optional vertical flip FIRST, then `rot` right-rotations (agent/player.py:300-305, 169-178).
RAZ_HD raz_bb bb_d4_apply(raz_bb x, int flip, int rot) {
    if (flip) x = bb_flip_vertical(x);
    if (rot & 2) x = bb_rotate180(x);
    if (rot & 1) x = bb_rotate90(x);
    return x;
}

// Where square s lands under the same transform (used to permute policy vectors: the value at
// source square s moves to bb_d4_square(s)).  Square = row*8+col.
RAZ_HD int bb_d4_square(int s, int flip, int rot) {
    int r = s >> 3, c = s & 7;
    if (flip) r = 7 - r;
    for (int i = 0; i < (rot & 3); ++i) {  // right rotation: (r,c) -> (c, 7-r)
        int nr = c, nc = 7 - r;
        r = nr;
        c = nc;
    }
    return r * 8 + c;
}

// One direction of the mobility fill, towards lower bit indices (lib/bitboard.py:95-104) and
// towards higher ones (:107-116).  The reference seeds with the enemy discs adjacent to an own disc
// and propagates five more single steps (a run of at most six flippable discs), then steps once
// more onto an empty square.  Because the edge mask limits every run of `e` along a line to six
// squares, propagating by 1, then 2, then 4 (parallel prefix) reaches exactly the same set for
// every input, in three steps instead of five.
RAZ_HD raz_bb bb_fill_down(raz_bb own, raz_bb e, int k) {
    raz_bb t = e & (own >> k);
    t |= e & (t >> k);
    raz_bb e2 = e & (e >> k);
    t |= e2 & (t >> (2 * k));
    raz_bb e4 = e2 & (e2 >> (2 * k));
    t |= e4 & (t >> (4 * k));
    return t >> k;
}
RAZ_HD raz_bb bb_fill_up(raz_bb own, raz_bb e, int k) {
    raz_bb t = e & (own << k);
    t |= e & (t << k);
    raz_bb e2 = e & (e << k);
    t |= e2 & (t << (2 * k));
    raz_bb e4 = e2 & (e2 << (2 * k));
    t |= e4 & (t << (4 * k));
    return t << k;
}

// Legal-move mask of `own` against `enemy`.  lib/bitboard.py:53-67.
RAZ_HD raz_bb bb_legal_moves(raz_bb own, raz_bb enemy) {
    const raz_bb blank = ~(own | enemy);
    const raz_bb e_lr = enemy & RAZ_MASK_LR, e_tb = enemy & RAZ_MASK_TB, e_in = enemy & RAZ_MASK_IN;
    raz_bb m = bb_fill_down(own, e_lr, 1) | bb_fill_up(own, e_lr, 1);
    m |= bb_fill_down(own, e_in, 9) | bb_fill_up(own, e_in, 9);
    m |= bb_fill_down(own, e_tb, 8) | bb_fill_up(own, e_tb, 8);
    m |= bb_fill_down(own, e_in, 7) | bb_fill_up(own, e_in, 7);
    return blank & m;
}

// Discs flipped along the four rays that run towards HIGHER bit indices from `pos`
// (down, right, down-left, down-right), by the carry-propagation trick of
// lib/bitboard.py:84-92: adding 1 to (e | ~ray) ripples through the contiguous enemy run that
// starts next to pos and lands on the first non-enemy square; if that square is ours the run is
// outflanked.  u64 wrap-around == the reference's big-int arithmetic under its `& own` clamp.
RAZ_HD raz_bb bb_flip_half(int pos, raz_bb own, raz_bb enemy) {
    const raz_bb e_lr = enemy & RAZ_MASK_LR;
    raz_bb flipped = 0, ray, out;
    ray = 0x0101010101010100ULL << pos;
    out = ray & ((enemy | ~ray) + 1) & own;
    flipped |= (out - (raz_bb)(out != 0)) & ray;
    ray = 0x00000000000000feULL << pos;
    out = ray & ((e_lr | ~ray) + 1) & own;
    flipped |= (out - (raz_bb)(out != 0)) & ray;
    ray = 0x0002040810204080ULL << pos;
    out = ray & ((e_lr | ~ray) + 1) & own;
    flipped |= (out - (raz_bb)(out != 0)) & ray;
    ray = 0x8040201008040200ULL << pos;
    out = ray & ((e_lr | ~ray) + 1) & own;
    flipped |= (out - (raz_bb)(out != 0)) & ray;
    return flipped;
}

// Flip mask for placing an `own` disc on `pos` (0..63).  The four rays towards lower indices are
// handled on the 180-degree-rotated board.  lib/bitboard.py:70-81.
RAZ_HD raz_bb bb_calc_flip(int pos, raz_bb own, raz_bb enemy) {
    raz_bb f1 = bb_flip_half(pos, own, enemy);
    raz_bb f2 = bb_flip_half(63 - pos, bb_rotate180(own), bb_rotate180(enemy));
    return f1 | bb_rotate180(f2);
}

// ---- game state machine (env/reversi_env.py) -------------------------------------------------
// player: 1 = black, 2 = white (Player enum, reversi_env.py:9).
// status: 0 = running; 1/2/3 = done with Winner black/white/draw (Winner enum, :11);
//         bit 0x10 set additionally when the game ended by an illegal (no-flip) move,
//         bit 0x20 when it ended by resignation.
#define RAZ_PLAYER_BLACK 1
#define RAZ_PLAYER_WHITE 2
#define RAZ_WIN_BLACK 1
#define RAZ_WIN_WHITE 2
#define RAZ_WIN_DRAW 3
#define RAZ_STATUS_WINNER_MASK 0x0f
#define RAZ_STATUS_ILLEGAL 0x10
#define RAZ_STATUS_RESIGNED 0x20
#define RAZ_ACTION_RESIGN 255  // the reference's `None` action (reversi_env.py:50-52)

typedef struct {
    raz_bb black, white;
    uint8_t player;  // side to move
    uint8_t status;
    raz_bb legal;  // legal moves of the side to move after the step (0 when done)
} raz_step_result;

// ReversiEnv.step(action) (reversi_env.py:42-74) on a running game.  `action` 0..63 or
// RAZ_ACTION_RESIGN.  Does NOT look at or require status==0 (the reference does not either).
RAZ_HD raz_step_result bb_env_step(raz_bb black, raz_bb white, int player, int action) {
    raz_step_result r;
    r.black = black;
    r.white = white;
    r.player = (uint8_t)player;
    r.legal = 0;
    const int other_wins = (player == RAZ_PLAYER_BLACK) ? RAZ_WIN_WHITE : RAZ_WIN_BLACK;
    if (action == RAZ_ACTION_RESIGN) {  // :50-52, 95-104
        r.status = (uint8_t)(other_wins | RAZ_STATUS_RESIGNED);
        return r;
    }
    raz_bb own = (player == RAZ_PLAYER_BLACK) ? black : white;
    raz_bb enemy = (player == RAZ_PLAYER_BLACK) ? white : black;
    raz_bb flipped = bb_calc_flip(action, own, enemy);
    if (flipped == 0) {  // :56-59, 90-93: mover loses, board untouched
        r.status = (uint8_t)(other_wins | RAZ_STATUS_ILLEGAL);
        return r;
    }
    own ^= flipped;
    own |= (raz_bb)1 << action;
    enemy ^= flipped;
    if (player == RAZ_PLAYER_BLACK) {
        r.black = own;
        r.white = enemy;
    } else {
        r.white = own;
        r.black = enemy;
    }
    raz_bb m = bb_legal_moves(enemy, own);  // :66
    if (m) {
        r.player = (uint8_t)(3 - player);
        r.status = 0;
        r.legal = m;
        return r;
    }
    m = bb_legal_moves(own, enemy);  // :68 (opponent passes)
    if (m) {
        r.status = 0;
        r.legal = m;
        return r;
    }
    int nb = bb_popcount(r.black), nw = bb_popcount(r.white);  // :76-85
    r.status = (uint8_t)(nb > nw ? RAZ_WIN_BLACK : (nb < nw ? RAZ_WIN_WHITE : RAZ_WIN_DRAW));
    return r;
}
