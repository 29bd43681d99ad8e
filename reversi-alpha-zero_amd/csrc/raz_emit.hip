// raz_emit.hip — the play_*.json text of finished games, written natively (host code).
//
// Reference: SelfPlayWorker.save_play_data (worker/self_play.py:180-194) extends its buffer with
// black.moves + white.moves and json.dump()s it (lib/data_helper.py:23-25); a ReversiPlayer's moves are
// the 8 symmetric rows [(own, enemy), list(policy64)] of every searched ply
// (add_data_to_move_buffer_with_8_symmetries, agent/player.py:166-179: flip in {F,T} x rot_right in 0..3,
// boards by flip_vertical / rotate90, policy by np.flipud / np.rot90(k=-rot)) with z appended by
// finish_game (:357-364; black rows get black_win, white rows -black_win, worker/self_play.py:219-231);
// the saved policy is calc_policy / calc_policy_by_tau_1 of the root visit counts (:132, 366-385).
//
// The device engine plays 4096 mini-net games in well under a second; building these rows as Python
// lists and json.dump()ing them took ~25 ms per game (226 KB of text), i.e. the worker was host-bound
// 100 to 1.  This emitter produces the SAME BYTES json.dump produces (", " separators, ints in decimal,
// floats as float.__repr__: the shortest string that round-trips) straight from the engine's records,
// formatting each ply's 64 policy values once and permuting the strings for the 8 symmetries.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "raz_bitboard.h"
#include "raz_engine.h"
#include "raz_internal.h"

namespace {

// float.__repr__ (CPython float_repr_style = 'short': David Gay's shortest round-trip digits, then
// format_float_short with code 'r').  The correctly rounded p-digit decimal (glibc's %.{p-1}e) is the
// p-digit decimal closest to x, so it round-trips iff ANY p-digit decimal does; and if fewer than 15
// digits suffice, the 15-digit rounding is that short decimal padded with zeros (normal doubles: their
// spacing is below 1e-15 relative).  Hence: first of 15 / 16 / 17 digits that round-trips, trailing zeros
// stripped; subnormals, whose spacing is coarse, search from one digit up.
int fmt_repr(double x, char* out) {
    if (!(fabs(x) <= 1.7976931348623157e308)) {   // float.__repr__ of non-finite values (json.dump writes NaN / Infinity)
        const char* t = x != x ? "NaN" : (x > 0 ? "Infinity" : "-Infinity");
        const int n = (int)strlen(t);
        memcpy(out, t, (size_t)n);
        return n;
    }
    if (x == 0.0) {
        const bool neg = signbit(x);
        memcpy(out, neg ? "-0.0" : "0.0", neg ? 4 : 3);
        return neg ? 4 : 3;
    }
    char buf[48];
    for (int prec = fabs(x) < 2.2250738585072014e-308 ? 1 : 15; prec <= 17; ++prec) {
        snprintf(buf, sizeof buf, "%.*e", prec - 1, x);
        if (prec == 17 || strtod(buf, nullptr) == x) break;
    }
    // buf = [-]d.ddd...e[+-]XX[X]
    const char* p = buf;
    char* o = out;
    if (*p == '-') *o++ = *p++;
    char digits[20];
    int nd = 0;
    for (; *p && *p != 'e'; ++p)
        if (*p >= '0' && *p <= '9') digits[nd++] = *p;
    while (nd > 1 && digits[nd - 1] == '0') --nd;
    const int exp10 = atoi(p + 1);
    const int decpt = exp10 + 1;
    if (decpt <= -4 || decpt > 16) {  // exponent notation: d[.ddd]e[+-]XX
        *o++ = digits[0];
        if (nd > 1) {
            *o++ = '.';
            memcpy(o, digits + 1, (size_t)(nd - 1));
            o += nd - 1;
        }
        o += sprintf(o, "e%c%02d", exp10 < 0 ? '-' : '+', exp10 < 0 ? -exp10 : exp10);
    } else if (decpt <= 0) {  // 0.000ddd
        *o++ = '0';
        *o++ = '.';
        for (int i = 0; i < -decpt; ++i) *o++ = '0';
        memcpy(o, digits, (size_t)nd);
        o += nd;
    } else if (decpt >= nd) {  // ddd000.0
        memcpy(o, digits, (size_t)nd);
        o += nd;
        for (int i = 0; i < decpt - nd; ++i) *o++ = '0';
        *o++ = '.';
        *o++ = '0';
    } else {  // dd.ddd
        memcpy(o, digits, (size_t)decpt);
        o += decpt;
        *o++ = '.';
        memcpy(o, digits + decpt, (size_t)(nd - decpt));
        o += nd - decpt;
    }
    return (int)(o - out);
}

// source index of every square of the policy after np.flipud (if flip) and np.rot90(k=-rot):
// flipud: F[r][c] = P[7-r][c];  one clockwise quarter turn: R[r][c] = M[7-c][r]
void policy_permutation(int flip, int rot, int perm[64]) {
    int cur[64], nxt[64];
    for (int i = 0; i < 64; ++i) cur[i] = i;
    if (flip) {
        for (int r = 0; r < 8; ++r)
            for (int c = 0; c < 8; ++c) nxt[r * 8 + c] = cur[(7 - r) * 8 + c];
        memcpy(cur, nxt, sizeof cur);
    }
    for (int k = 0; k < rot; ++k) {
        for (int r = 0; r < 8; ++r)
            for (int c = 0; c < 8; ++c) nxt[r * 8 + c] = cur[(7 - c) * 8 + r];
        memcpy(cur, nxt, sizeof cur);
    }
    memcpy(perm, cur, sizeof cur);
}

}  // namespace

extern "C" int raz_format_float_repr(double x, char* out32) {
    if (!out32) return raz_fail(RAZ_EINVAL, "raz_format_float_repr: NULL out");
    const int n = fmt_repr(x, out32);
    out32[n] = 0;
    return n;
}

extern "C" long long raz_emit_game_rows_json(const void* headers, const uint32_t* root_n, int n_plies, int winner,
                                             int change_tau_turn, int save_policy_of_tau_1, char* out, size_t cap,
                                             int* n_rows) {
    if (!headers || !root_n || !out || n_plies < 0) return raz_fail(RAZ_EINVAL, "raz_emit_game_rows_json: bad argument");
    const raz_ply_header* H = (const raz_ply_header*)headers;
    struct Perms {   // built once; a function-local static's initialisation is thread-safe (C++11) - callers are host threads
        int p[8][64];
        Perms() {
            for (int f = 0; f < 2; ++f)
                for (int r = 0; r < 4; ++r) policy_permutation(f, r, p[f * 4 + r]);
        }
    };
    static const Perms perms;
    const int (*perm)[64] = perms.p;
    const int black_win = winner == RAZ_WIN_BLACK ? 1 : (winner == RAZ_WIN_WHITE ? -1 : 0);
    size_t pos = 0;
    int rows = 0;
    for (int side = 1; side <= 2; ++side) {  // black.moves + white.moves (worker/self_play.py:183)
        const int z = side == 1 ? black_win : -black_win;
        for (int j = 0; j < n_plies; ++j) {
            const raz_ply_header& h = H[j];
            if (h.player != side || !h.has_row) continue;
            const uint32_t* n = root_n + (size_t)j * 64;
            // the saved policy (agent/player.py:132, 366-385), each value formatted once
            char txt[64][28];
            int len[64];
            if (save_policy_of_tau_1 || (int)h.turn < change_tau_turn) {
                double sum = 0.0;  // np.sum of integers below 2^53: exact in any order
                for (int i = 0; i < 64; ++i) sum += (double)n[i];
                // (sum == 0 cannot come from a searched ply; numpy would give nan = "NaN" in the reference's json.dump)
                for (int i = 0; i < 64; ++i) len[i] = fmt_repr((double)n[i] / sum, txt[i]);
            } else {
                int am = 0;
                for (int i = 1; i < 64; ++i)
                    if (n[i] > n[am]) am = i;  // np.argmax: first maximum
                for (int i = 0; i < 64; ++i) {
                    memcpy(txt[i], i == am ? "1.0" : "0.0", 3);
                    len[i] = 3;
                }
            }
            for (int f = 0; f < 2; ++f) {
                for (int r = 0; r < 4; ++r) {
                    raz_bb o = h.own, e = h.enemy;
                    if (f) {
                        o = bb_flip_vertical(o);
                        e = bb_flip_vertical(e);
                    }
                    for (int k = 0; k < r; ++k) {
                        o = bb_rotate90(o);
                        e = bb_rotate90(e);
                    }
                    if (pos + 64 * 30 + 96 > cap) return raz_fail(RAZ_ENOMEM, "raz_emit_game_rows_json: output buffer too small");
                    char* w = out + pos;
                    if (rows) {
                        *w++ = ',';
                        *w++ = ' ';
                    }
                    w += sprintf(w, "[[%llu, %llu], [", o, e);
                    const int* pm = perm[f * 4 + r];
                    for (int i = 0; i < 64; ++i) {
                        if (i) {
                            *w++ = ',';
                            *w++ = ' ';
                        }
                        memcpy(w, txt[pm[i]], (size_t)len[pm[i]]);
                        w += len[pm[i]];
                    }
                    w += sprintf(w, "], %d]", z);
                    pos = (size_t)(w - out);
                    ++rows;
                }
            }
        }
    }
    if (n_rows) *n_rows = rows;
    return (long long)pos;
}
