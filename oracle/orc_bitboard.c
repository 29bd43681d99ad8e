/* orc_bitboard.c — CPU oracle: lib/bitboard.py and env/reversi_env.py restated in plain C.
 * TEST INFRASTRUCTURE (see orc.h).  Deliberately follows the Python line by line (lists of masks,
 * rotate180 as two rotate90s, loop popcount) rather than the device code's formulation, so that
 * the two implementations fail independently.
 */
#include "orc.h"

/* lib/bitboard.py:132-133 — bin(x).count('1') */
int orc_bit_count(u64 x) {
    int c = 0;
    while (x) {
        c += (int)(x & 1);
        x >>= 1;
    }
    return c;
}

/* lib/bitboard.py:136-138 */
void orc_bit_to_array(u64 x, int size, uint8_t* out) {
    for (int i = 0; i < size; ++i) out[i] = (uint8_t)((x >> i) & 1);
}

/* lib/bitboard.py:119-125 */
u64 orc_flip_vertical(u64 x) {
    const u64 k1 = 0x00FF00FF00FF00FFULL, k2 = 0x0000FFFF0000FFFFULL;
    x = ((x >> 8) & k1) | ((x & k1) << 8);
    x = ((x >> 16) & k2) | ((x & k2) << 16);
    x = (x >> 32) | (x << 32);
    return x;
}

/* lib/bitboard.py:141-151 */
u64 orc_flip_diag_a1h8(u64 x) {
    const u64 k1 = 0x5500550055005500ULL, k2 = 0x3333000033330000ULL, k4 = 0x0f0f0f0f00000000ULL;
    u64 t;
    t = k4 & (x ^ (x << 28));
    x ^= t ^ (t >> 28);
    t = k2 & (x ^ (x << 14));
    x ^= t ^ (t >> 14);
    t = k1 & (x ^ (x << 7));
    x ^= t ^ (t >> 7);
    return x;
}

/* lib/bitboard.py:154-155 */
u64 orc_rotate90(u64 x) { return orc_flip_diag_a1h8(orc_flip_vertical(x)); }
/* lib/bitboard.py:158-159 */
u64 orc_rotate180(u64 x) { return orc_rotate90(orc_rotate90(x)); }

/* lib/bitboard.py:95-104 */
static u64 search_offset_left(u64 own, u64 enemy, u64 mask, int offset) {
    u64 e = enemy & mask;
    u64 blank = ~(own | enemy);
    u64 t = e & (own >> offset);
    for (int i = 0; i < 5; ++i) t |= e & (t >> offset);
    return blank & (t >> offset);
}

/* lib/bitboard.py:107-116.  Python's unbounded `<<` never loses high bits, but every use is
 * immediately AND-ed with e (< 2^64) or blank's low 64 bits, so u64 truncation is equivalent. */
static u64 search_offset_right(u64 own, u64 enemy, u64 mask, int offset) {
    u64 e = enemy & mask;
    u64 blank = ~(own | enemy);
    u64 t = e & (own << offset);
    for (int i = 0; i < 5; ++i) t |= e & (t << offset);
    return blank & (t << offset);
}

/* lib/bitboard.py:53-67 */
u64 orc_find_correct_moves(u64 own, u64 enemy) {
    const u64 left_right_mask = 0x7e7e7e7e7e7e7e7eULL;
    const u64 top_bottom_mask = 0x00ffffffffffff00ULL;
    const u64 mask = left_right_mask & top_bottom_mask;
    u64 mobility = 0;
    mobility |= search_offset_left(own, enemy, left_right_mask, 1);
    mobility |= search_offset_left(own, enemy, mask, 9);
    mobility |= search_offset_left(own, enemy, top_bottom_mask, 8);
    mobility |= search_offset_left(own, enemy, mask, 7);
    mobility |= search_offset_right(own, enemy, left_right_mask, 1);
    mobility |= search_offset_right(own, enemy, mask, 9);
    mobility |= search_offset_right(own, enemy, top_bottom_mask, 8);
    mobility |= search_offset_right(own, enemy, mask, 7);
    return mobility;
}

/* lib/bitboard.py:84-92.  `(e | ~mask) + 1` and `outflank - (outflank != 0)` are two's-complement
 * identities on Python's big ints; the results are clamped by `mask &` / `& own` (both < 2^64),
 * so mod-2^64 arithmetic gives the same bits (the Cython copy alt/bitboard_cython.pyx:32-40 relies
 * on the same fact). */
static u64 calc_flip_half(int pos, u64 own, u64 enemy) {
    u64 el[4] = {enemy, enemy & 0x7e7e7e7e7e7e7e7eULL, enemy & 0x7e7e7e7e7e7e7e7eULL,
                 enemy & 0x7e7e7e7e7e7e7e7eULL};
    u64 masks[4] = {0x0101010101010100ULL, 0x00000000000000feULL, 0x0002040810204080ULL,
                    0x8040201008040200ULL};
    u64 flipped = 0;
    for (int i = 0; i < 4; ++i) {
        u64 mask = masks[i] << pos; /* b64(m << pos) */
        u64 outflank = mask & ((el[i] | ~mask) + 1) & own;
        flipped |= (outflank - (u64)(outflank != 0)) & mask;
    }
    return flipped;
}

/* lib/bitboard.py:70-81 (the assert on pos is the caller's job here) */
u64 orc_calc_flip(int pos, u64 own, u64 enemy) {
    u64 f1 = calc_flip_half(pos, own, enemy);
    u64 f2 = calc_flip_half(63 - pos, orc_rotate180(own), orc_rotate180(enemy));
    return f1 | orc_rotate180(f2);
}

/* env/reversi_env.py:133-143 — Board(): `black or default` */
static void board_set(orc_env* e, u64 black, u64 white) {
    e->black = black ? black : (((u64)0x10 << 24) | ((u64)0x08 << 32));
    e->white = white ? white : (((u64)0x08 << 24) | ((u64)0x10 << 32));
}

/* env/reversi_env.py:26-32 */
void orc_env_reset(orc_env* e) {
    board_set(e, 0, 0);
    e->next_player = 1;
    e->turn = 0;
    e->done = 0;
    e->winner = 0;
    e->ended_illegal = e->ended_resign = 0;
}

/* env/reversi_env.py:34-40 */
void orc_env_update(orc_env* e, u64 black, u64 white, int next_player) {
    board_set(e, black, white);
    e->next_player = next_player;
    e->turn = orc_bit_count(e->black) + orc_bit_count(e->white) - 4;
    e->done = 0;
    e->winner = 0;
    e->ended_illegal = e->ended_resign = 0;
}

/* env/reversi_env.py:76-85 */
static void game_over(orc_env* e) {
    e->done = 1;
    if (e->winner == 0) {
        int b = orc_bit_count(e->black), w = orc_bit_count(e->white);
        e->winner = b > w ? 1 : (b < w ? 2 : 3);
    }
}

/* env/reversi_env.py:99-104 */
static void win_another_player(orc_env* e) { e->winner = (e->next_player == 1) ? 2 : 1; }

/* env/reversi_env.py:42-74 */
void orc_env_step(orc_env* e, int action) {
    if (action < 0) { /* None -> _resigned (:95-97) */
        win_another_player(e);
        game_over(e);
        e->ended_resign = 1;
        return;
    }
    u64 own = e->next_player == 1 ? e->black : e->white; /* :106-111 */
    u64 enemy = e->next_player == 1 ? e->white : e->black;
    u64 flipped = orc_calc_flip(action, own, enemy);
    if (orc_bit_count(flipped) == 0) { /* :57-59, 90-93 */
        win_another_player(e);
        game_over(e);
        e->ended_illegal = 1;
        return;
    }
    own ^= flipped;
    own |= (u64)1 << action;
    enemy ^= flipped;
    if (e->next_player == 1) { /* :113-117 */
        e->black = own;
        e->white = enemy;
    } else {
        e->white = own;
        e->black = enemy;
    }
    e->turn += 1;
    if (orc_bit_count(orc_find_correct_moves(enemy, own)) > 0)
        e->next_player = 3 - e->next_player; /* :66-67 */
    else if (orc_bit_count(orc_find_correct_moves(own, enemy)) > 0) {
        /* :68-69 pass */
    } else
        game_over(e); /* :70-71 */
}

/* ---- array helpers so tests can check millions of boards quickly (plain loops) ---- */
void orc_find_correct_moves_n(const u64* own, const u64* enemy, u64* out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = orc_find_correct_moves(own[i], enemy[i]);
}
void orc_calc_flip_n(const uint8_t* pos, const u64* own, const u64* enemy, u64* out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = pos[i] < 64 ? orc_calc_flip(pos[i], own[i], enemy[i]) : 0;
}
/* One ReversiEnv.step per game on SoA arrays; same conventions as include/raz.h raz_step_batch:
 * status 0 running, else winner | 0x10 illegal | 0x20 resigned; done games untouched, legal = 0. */
void orc_step_n(u64* black, u64* white, uint8_t* player, uint8_t* status, u64* legal,
                const uint8_t* action, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        if (status[i]) {
            legal[i] = 0;
            continue;
        }
        orc_env e;
        e.black = black[i];
        e.white = white[i];
        e.next_player = player[i];
        e.turn = 0;
        e.done = 0;
        e.winner = 0;
        e.ended_illegal = e.ended_resign = 0;
        orc_env_step(&e, action[i] == 255 ? -1 : action[i]);
        black[i] = e.black;
        white[i] = e.white;
        player[i] = (uint8_t)e.next_player;
        status[i] = (uint8_t)(e.done ? (e.winner | (e.ended_illegal ? 0x10 : 0) | (e.ended_resign ? 0x20 : 0)) : 0);
        if (e.done)
            legal[i] = 0;
        else
            legal[i] = e.next_player == 1 ? orc_find_correct_moves(e.black, e.white)
                                          : orc_find_correct_moves(e.white, e.black);
    }
}
