// raz_bitboard_valu.h — the mobility / flip / step primitives of raz_bitboard.h re-shaped for the VECTOR ALU of gfx950.
//
// raz_bitboard.h states the reference's integer ops (lib/bitboard.py:53-116, env/reversi_env.py:42-85) on u64 values.
// That is the right shape for the tree kernel, where a board is wave-uniform and the compiler keeps it in SGPRs (the
// scalar ALU has native 64-bit logic and shifts).  In the sweep kernels (raz_sweep.hip) a board is per-LANE, every u64
// logic op becomes two 32-bit VALU instructions, and the kernels are bound by VALU issue, not by HBM.
// Same results for ALL 2^64 x 2^64 inputs, fewer vector instructions:
//   * three-input logic - (a & b) | c, a & b & c, a & ~(b | c) - is ONE instruction per half (v_bitop3_b32, a gfx950
//     instruction); the compiler does not form it from 64-bit source because it splits 64-bit logic after instruction
//     selection, so the helpers below spell it out on the halves;
//   * shifts stay single 64-bit instructions (v_lshrrev_b64 / v_lshlrev_b64) although only halves of their results are
//     consumed (left alone the compiler would split each into v_alignbit_b32 + a 32-bit shift);
//   * the parallel-prefix fill reuses its stride-2 mask twice (runs are at most 6 long: 1 + 1 + 2 + 2) instead of
//     building a stride-4 mask;
//   * the two horizontal directions use carry propagation: adding the seeds (own << 1) & e to the masked enemy row
//     ripples through each seeded run and lands on the square behind it - one 64-bit add instead of a prefix fill; the
//     direction towards lower bits is the same computation on the bit-reversed board (v_bfrev_b32);
//   * bbv_env_step_first reports "the opponent cannot move" instead of computing the mover's own mobility in line, so
//     that a kernel can do that second, rare computation only for the boards that need it (k_step).
// The host build (plain C++) is checked against raz_bitboard.h by tests/native/bbv_check.cpp; the device build by the
// sweep parity tests (tests/test_sweep_gpu.py), which compare the kernels with the oracle.
#pragma once
#include "raz_bitboard.h"

RAZ_HD uint32_t bbv_lo(raz_bb x) { return (uint32_t)x; }
RAZ_HD uint32_t bbv_hi(raz_bb x) { return (uint32_t)(x >> 32); }
RAZ_HD raz_bb bbv_join(uint32_t lo, uint32_t hi) { return ((raz_bb)hi << 32) | lo; }

// v_bitop3_b32 evaluates the boolean function whose truth table is f(0xF0, 0xCC, 0xAA).
#if defined(__HIP_DEVICE_COMPILE__)
#define RAZ_BBV_OP3(name, table, expr)                                                               \
    RAZ_HD raz_bb name(raz_bb a, raz_bb b, raz_bb c) {                                               \
        return bbv_join(__builtin_amdgcn_bitop3_b32(bbv_lo(a), bbv_lo(b), bbv_lo(c), table),         \
                        __builtin_amdgcn_bitop3_b32(bbv_hi(a), bbv_hi(b), bbv_hi(c), table));        \
    }
#else
#define RAZ_BBV_OP3(name, table, expr) \
    RAZ_HD raz_bb name(raz_bb a, raz_bb b, raz_bb c) { return expr; }
#endif
RAZ_BBV_OP3(bbv_and_or, 0xEA, (a & b) | c)        // (0xF0 & 0xCC) | 0xAA
RAZ_BBV_OP3(bbv_and3, 0x80, a & b & c)            // 0xF0 & 0xCC & 0xAA
RAZ_BBV_OP3(bbv_and_nor, 0x10, a & ~(b | c))      // 0xF0 & ~(0xCC | 0xAA)
RAZ_BBV_OP3(bbv_or3, 0xFE, a | b | c)             // 0xF0 | 0xCC | 0xAA

template <int K> RAZ_HD raz_bb bbv_shr(raz_bb x) {
#if defined(__HIP_DEVICE_COMPILE__)
    raz_bb r;
    asm("v_lshrrev_b64 %0, %1, %2" : "=v"(r) : "n"(K), "v"(x));
    return r;
#else
    return x >> K;
#endif
}
template <int K> RAZ_HD raz_bb bbv_shl(raz_bb x) {
#if defined(__HIP_DEVICE_COMPILE__)
    raz_bb r;
    asm("v_lshlrev_b64 %0, %1, %2" : "=v"(r) : "n"(K), "v"(x));
    return r;
#else
    return x << K;
#endif
}

// One direction of the mobility fill (bb_fill_down / bb_fill_up): seed, then reach 2, 4, 6 squares.
template <int K> RAZ_HD raz_bb bbv_fill_down(raz_bb own, raz_bb e) {
    raz_bb t = e & bbv_shr<K>(own);
    t = bbv_and_or(e, bbv_shr<K>(t), t);
    const raz_bb e2 = e & bbv_shr<K>(e);
    t = bbv_and_or(e2, bbv_shr<2 * K>(t), t);
    t = bbv_and_or(e2, bbv_shr<2 * K>(t), t);
    return bbv_shr<K>(t);
}
template <int K> RAZ_HD raz_bb bbv_fill_up(raz_bb own, raz_bb e) {
    raz_bb t = e & bbv_shl<K>(own);
    t = bbv_and_or(e, bbv_shl<K>(t), t);
    const raz_bb e2 = e & bbv_shl<K>(e);
    t = bbv_and_or(e2, bbv_shl<2 * K>(t), t);
    t = bbv_and_or(e2, bbv_shl<2 * K>(t), t);
    return bbv_shl<K>(t);
}

// The squares behind the runs of e_lr (enemy & columns 1..6) that start right above an own disc, along a row towards
// higher bit indices.  A seed is an enemy disc whose lower neighbour is ours; the add carries from the lowest seed of
// a run through the run's top and sets the first bit that is not in e_lr (column 7 at the latest: e_lr has no column 0
// or 7, so a carry never leaves its row).  Bits the add leaves inside a run are masked off - the reference's fill also
// produces them (t << 1 inside the run) and drops them with `blank &`.
RAZ_HD raz_bb bbv_row_up(raz_bb own, raz_bb e_lr) {
    const raz_bb seeds = (own << 1) & e_lr;
    return (e_lr + seeds) & ~e_lr;
}

// Legal-move mask of `own` against `enemy`: == bb_legal_moves for every input.
RAZ_HD raz_bb bbv_legal_moves(raz_bb own, raz_bb enemy) {
    const raz_bb e_lr = enemy & RAZ_MASK_LR, e_tb = enemy & RAZ_MASK_TB, e_in = e_lr & RAZ_MASK_TB;
    raz_bb m = bbv_row_up(own, e_lr) | bb_rotate180(bbv_row_up(bb_rotate180(own), bb_rotate180(enemy) & RAZ_MASK_LR));   // (the mask is symmetric)
    m = bbv_or3(m, bbv_fill_down<9>(own, e_in), bbv_fill_up<9>(own, e_in));
    m = bbv_or3(m, bbv_fill_down<8>(own, e_tb), bbv_fill_up<8>(own, e_tb));
    m = bbv_or3(m, bbv_fill_down<7>(own, e_in), bbv_fill_up<7>(own, e_in));
    return bbv_and_nor(m, own, enemy);
}

// bb_flip_half / bb_calc_flip (lib/bitboard.py:70-92) with the three-input forms.
RAZ_HD raz_bb bbv_flip_ray(raz_bb ray, raz_bb e, raz_bb own, raz_bb flipped) {
    const raz_bb out = bbv_and3(ray, (e | ~ray) + 1, own);
    return bbv_and_or(out - (raz_bb)(out != 0), ray, flipped);
}
RAZ_HD raz_bb bbv_flip_half(int pos, raz_bb own, raz_bb enemy) {
    const raz_bb e_lr = enemy & RAZ_MASK_LR;
    raz_bb f = bbv_flip_ray(0x0101010101010100ULL << pos, enemy, own, 0);
    f = bbv_flip_ray(0x00000000000000feULL << pos, e_lr, own, f);
    f = bbv_flip_ray(0x0002040810204080ULL << pos, e_lr, own, f);
    return bbv_flip_ray(0x8040201008040200ULL << pos, e_lr, own, f);
}
RAZ_HD raz_bb bbv_calc_flip(int pos, raz_bb own, raz_bb enemy) {
    return bbv_flip_half(pos, own, enemy) | bb_rotate180(bbv_flip_half(63 - pos, bb_rotate180(own), bb_rotate180(enemy)));
}

// bb_env_step without the in-line second mobility computation: when the opponent has no move after the step,
// `status` is left at RAZ_STEP_OPP_STUCK and the caller finishes with bbv_env_step_finish (which needs one more
// bbv_legal_moves).  Everything else - resign, illegal move, flips, side switch - as bb_env_step.
#define RAZ_STEP_OPP_STUCK 0x80
RAZ_HD raz_step_result bbv_env_step_first(raz_bb black, raz_bb white, int player, int action) {
    raz_step_result r;
    r.black = black;
    r.white = white;
    r.player = (uint8_t)player;
    r.legal = 0;
    const int other_wins = (player == RAZ_PLAYER_BLACK) ? RAZ_WIN_WHITE : RAZ_WIN_BLACK;
    if (action == RAZ_ACTION_RESIGN) {
        r.status = (uint8_t)(other_wins | RAZ_STATUS_RESIGNED);
        return r;
    }
    raz_bb own = (player == RAZ_PLAYER_BLACK) ? black : white;
    raz_bb enemy = (player == RAZ_PLAYER_BLACK) ? white : black;
    const raz_bb flipped = bbv_calc_flip(action, own, enemy);
    if (flipped == 0) {
        r.status = (uint8_t)(other_wins | RAZ_STATUS_ILLEGAL);
        return r;
    }
    own ^= flipped;
    own |= (raz_bb)1 << action;
    enemy ^= flipped;
    r.black = (player == RAZ_PLAYER_BLACK) ? own : enemy;
    r.white = (player == RAZ_PLAYER_BLACK) ? enemy : own;
    const raz_bb m = bbv_legal_moves(enemy, own);
    if (m) {
        r.player = (uint8_t)(3 - player);
        r.status = 0;
        r.legal = m;
    } else {
        r.status = RAZ_STEP_OPP_STUCK;
    }
    return r;
}
// env/reversi_env.py:68-85 for a board bbv_env_step_first left at RAZ_STEP_OPP_STUCK: the mover moves again if it
// can, else the game is over and the discs are counted.
RAZ_HD void bbv_env_step_finish(raz_step_result& r) {
    const raz_bb own = (r.player == RAZ_PLAYER_BLACK) ? r.black : r.white;
    const raz_bb enemy = (r.player == RAZ_PLAYER_BLACK) ? r.white : r.black;
    const raz_bb m = bbv_legal_moves(own, enemy);
    if (m) {
        r.status = 0;
        r.legal = m;
        return;
    }
    const int nb = bb_popcount(r.black), nw = bb_popcount(r.white);
    r.status = (uint8_t)(nb > nw ? RAZ_WIN_BLACK : (nb < nw ? RAZ_WIN_WHITE : RAZ_WIN_DRAW));
}
