// raz_sweep_sliced.h — the sweep kernels' arithmetic in BIT-SLICED form: one lane works on 32 boards at once.
//
// raz_sweep.hip's one-board-per-lane kernels are bound by vector-ALU issue, not by HBM (k_step 344, k_legal_moves 134 instructions
// per board: every u64 shift / and / or of lib/bitboard.py:53-116 is two or three 32-bit vector instructions, and a fill step is a
// dependent shift + and + or).  Here a lane holds 32 boards TRANSPOSED: word S[s] (s = 0..63, bit 0 = top-left as in
// lib/bitboard.py:10-17) carries square s of the lane's 32 boards, one board per bit.  In that form
//   * a shift by one square in any of the 8 directions is a different register name - free;
//   * the reference's fills  t |= e & (t >> k)  (find_correct_moves' search_offset_left / right, :95-116) become a walk along the line,
//     t[s] = e[s] & (own[s - k] | t[s - k]): ONE three-input instruction (v_bitop3_b32) per square and direction for 32 boards;
//   * the edge masks of the reference (0x7e7e..., 0x00ffffffffffff00 and their AND) are compile-time facts about s: a masked-out
//     square simply has no instruction.
// find_correct_moves costs 656 instructions per 32 boards (20.5 per board instead of 134), calc_flip ~26 per board instead of 111.
// The price is the 32 x 32 bit transposes between the memory layout (u64 per board) and the sliced one: 256 instructions per 32 words
// (byte permutes for the 16- and 8-bit stages, shift + bitop3 select for 4 / 2 / 1), i.e. 16 per board and bitboard.
// Which 32 boards a lane holds does not matter as long as loads and stores agree: a wave takes a SUPERBLOCK of 2048 consecutive boards
// as 16 rows of 128, lane l loading boards 2l, 2l + 1 of every row (global_load_dwordx4, each instruction one contiguous KiB) - board
// k = 2 * row + j of the lane.
// Results equal raz_bitboard.h's (== the reference's) for ALL inputs, overlapping own / enemy garbage included: the walks below are
// the same recurrences square by square (see the notes at each function); tests/test_sweep_emu.py and tests/test_sweep_gpu.py compare
// the kernels with the CPU oracle.
#pragma once
#include <stdint.h>
#include <utility>
#include "raz_bitboard.h"

namespace sl {

#define RAZ_SL __device__ __forceinline__
// keeps the instruction scheduler from mixing two phases of a kernel (it would otherwise start the next phase's independent work early
// and hold both phases' registers at once)
// a pointer every lane holds the same value of, handed to the compiler as a scalar it cannot fold the lane's offset into: accesses
// p[lane + constant] then take ONE vector register (the lane's 32-bit offset) for all the rows of all the streams, instead of a 64-bit
// address pair per 4 KB of every stream
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long v = (unsigned long long)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    typedef T __attribute__((address_space(1))) * global_ptr;   // (device memory, as the kernel's arguments are: global_ not flat_ accesses)
    return (T*)(global_ptr)(((unsigned long long)hi << 32) | lo);
#else
    return p;
#endif
}

// the lane's number in a ONE-WAVE workgroup, made anew (two instructions) where a phase needs it instead of held in a register
// through the phases before
__device__ __forceinline__ unsigned lane_again() {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned l;   // (volatile: made where this stands, every time - not once before the loop and kept, which is what it is here to avoid)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
#elif defined(RAZ_WAVE_EMU)
    return threadIdx.x;
#else
    return 0u;   // (plain host harnesses of the arithmetic: no lanes)
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
#define RAZ_SL_PHASE() __builtin_amdgcn_sched_barrier(0)
#define RAZ_SL_PIN(x) asm volatile("" : "+v"(x))   // the value is made where this stands: the optimiser may not sink its computation to a later use
#else
#define RAZ_SL_PHASE() ((void)0)
#define RAZ_SL_PIN(x) ((void)0)
#endif

template <class F, int... I>
RAZ_SL void sfor_(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): every index a compile-time constant
template <int N, class F>
RAZ_SL void sfor(F&& f) {
    sfor_(f, std::make_integer_sequence<int, N>{});
}

// v_bitop3_b32: the boolean function of three words whose truth table is T = f(0xF0, 0xCC, 0xAA)
template <int T>
RAZ_SL uint32_t op3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, T);
#else
    uint32_t r = 0;
    if (T & 0x80) r |= a & b & c;
    if (T & 0x40) r |= a & b & ~c;
    if (T & 0x20) r |= a & ~b & c;
    if (T & 0x10) r |= a & ~b & ~c;
    if (T & 0x08) r |= ~a & b & c;
    if (T & 0x04) r |= ~a & b & ~c;
    if (T & 0x02) r |= ~a & ~b & c;
    if (T & 0x01) r |= ~a & ~b & ~c;
    return r;
#endif
}
constexpr int kAndOr = 0xE0;    // a & (b | c)
constexpr int kOr3 = 0xFE;      // a | b | c
constexpr int kAndNor = 0x10;   // a & ~(b | c)
constexpr int kSelect = 0xCA;   // a ? b : c  (bitwise)
constexpr int kOrAndN = 0xF4;   // a | (b & ~c)
constexpr int kAndN3 = 0x20;    // a & ~b & c

// bytes of {hi, lo} picked by the selector (v_perm_b32: selector byte 0-3 = byte of lo, 4-7 = byte of hi)
template <uint32_t SEL>
RAZ_SL uint32_t perm(uint32_t hi, uint32_t lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, SEL);
#else
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) r |= (uint32_t)((v >> (8 * ((SEL >> (8 * i)) & 7))) & 0xff) << (8 * i);
    return r;
#endif
}

// ---- 32 x 32 bit transpose of a[OFF .. OFF + 31]: afterwards bit k of a[OFF + i] is what bit i of a[OFF + k] was.  Five butterfly
// stages; stage J exchanges, for every pair of rows (k, k | J), the J-wide column blocks: new a[k] = (a[k] & m) | ((a[k|J] & m) << J),
// new a[k|J] = ((a[k] >> J) & m) | (a[k|J] & ~m), m = the columns whose bit J is clear.  An involution.
template <int J, int OFF>
RAZ_SL void tstage(uint32_t (&a)[64]) {
    constexpr uint32_t m = J == 16 ? 0x0000FFFFu : J == 8 ? 0x00FF00FFu : J == 4 ? 0x0F0F0F0Fu : J == 2 ? 0x33333333u : 0x55555555u;
    sfor<16>([&](auto g_) __attribute__((always_inline)) {
        constexpr int g = decltype(g_)::value;
        constexpr int k = OFF + (((g & ~(J - 1)) << 1) | (g & (J - 1)));
        const uint32_t lo = a[k], hi = a[k + J];
        if constexpr (J == 16) {
            a[k] = perm<0x05040100u>(hi, lo);
            a[k + J] = perm<0x07060302u>(hi, lo);
        } else if constexpr (J == 8) {
            a[k] = perm<0x06020400u>(hi, lo);
            a[k + J] = perm<0x07030501u>(hi, lo);
        } else {
            a[k] = op3<kSelect>(m, lo, hi << J);
            a[k + J] = op3<kSelect>(m, lo >> J, hi);
        }
    });
}
template <int OFF>
RAZ_SL void transpose32(uint32_t (&a)[64]) {
    tstage<16, OFF>(a);
    tstage<8, OFF>(a);
    tstage<4, OFF>(a);
    tstage<2, OFF>(a);
    tstage<1, OFF>(a);
}
// a[k] = low word, a[32 + k] = high word of board k  <->  a[s] = square s of the 32 boards
RAZ_SL void transpose_boards(uint32_t (&a)[64]) {
    transpose32<0>(a);
    transpose32<32>(a);
}

// ---- geometry: square s = 8 * row + col; direction D = 0..7
constexpr int kDr[8] = {0, 0, 1, -1, 1, -1, 1, -1};
constexpr int kDc[8] = {1, -1, 0, 0, 1, -1, -1, 1};   // D ^ 1 is the opposite direction
constexpr bool on_board(int r, int c) { return r >= 0 && r < 8 && c >= 0 && c < 8; }
// the reference's mask for the direction (lib/bitboard.py:57-66: 0x7e7e7e7e7e7e7e7e for the row direction, 0x00ffffffffffff00 for the
// column direction, their AND for the diagonals): the squares an enemy run may pass THROUGH
constexpr bool passable(int d, int s) {
    const int r = s >> 3, c = s & 7;
    return (kDc[d] == 0 || (c >= 1 && c <= 6)) && (kDr[d] == 0 || (r >= 1 && r <= 6));
}
constexpr int pred(int d, int s) {   // the square the walk in direction d comes from, or -1
    const int r = (s >> 3) - kDr[d], c = (s & 7) - kDc[d];
    return on_board(r, c) ? 8 * r + c : -1;
}
// t_d[s] can be non-zero: s is passable and has a predecessor
constexpr bool has_t(int d, int s) { return s >= 0 && passable(d, s) && pred(d, s) >= 0; }
// the squares in the order a walk in direction d must visit them (predecessor first)
constexpr int walk_square(int d, int i) { return (8 * kDr[d] + kDc[d]) > 0 ? i : 63 - i; }

// ---- lines.  Direction families F = 0 (rows), 1 (columns), 2 (diagonals, down-right), 3 (anti-diagonals, down-left); a line is named by
// its first square in the family's forward direction D = 2 F (the square without a predecessor).  Square i of the line: line_sq.
// The reference's masks say exactly "a run passes through INTERIOR squares of its line only" (passable(d, s) <=> s is neither end of
// its line in direction d: for a diagonal both coordinates of an interior square are in 1..6).
constexpr int line_sq(int d, int start, int i) {
    const int r = (start >> 3) + i * kDr[d], c = (start & 7) + i * kDc[d];
    return on_board(r, c) ? 8 * r + c : -1;
}
constexpr int line_len(int d, int start) {
    int n = 0;
    while (n < 8 && line_sq(d, start, n) >= 0) ++n;
    return n;
}
constexpr bool line_start(int d, int s) { return pred(d, s) < 0; }

// ---- find_correct_moves (lib/bitboard.py:53-67) for 32 boards.  Per direction the reference computes t = e_masked & shift(own), five
// times t |= e_masked & shift(t), and returns blank & shift(t).  Along a line q_0 .. q_(n-1) that is, forwards, t[i] = e[q_i] & (own[q_(i-1)]
// | t[i-1]) for the interior squares i = 1 .. n-2 (six steps reach every run a line of eight can hold) and a move on q_i where t[i-1] is
// set and q_i is blank; backwards the same from the other end.  own / enemy may overlap: seeds look at own only, runs at enemy only,
// blank = ~(own | enemy) - as in the reference.  RAW: L is not masked with blank yet.
// MODE 0: L[q] = contributions (the first family: it writes every square); 1: L[q] |= contributions; 2: L[q] |= contributions & mask.
template <int D, int S0, int MODE>
RAZ_SL void mobility_line(const uint32_t (&o)[64], const uint32_t (&e)[64], uint32_t (&L)[64], uint32_t mask) {
    constexpr int n = line_len(D, S0);
    if constexpr (n >= 3) {
        uint32_t tf[8], tb[8];
        sfor<n - 2>([&](auto j_) __attribute__((always_inline)) {
            constexpr int i = 1 + decltype(j_)::value, q = line_sq(D, S0, i), qp = line_sq(D, S0, i - 1);
            if constexpr (i == 1) tf[i] = e[q] & o[qp];
            else tf[i] = op3<kAndOr>(e[q], o[qp], tf[i - 1]);
        });
        sfor<n - 2>([&](auto j_) __attribute__((always_inline)) {
            constexpr int i = n - 2 - decltype(j_)::value, q = line_sq(D, S0, i), qn = line_sq(D, S0, i + 1);
            if constexpr (i == n - 2) tb[i] = e[q] & o[qn];
            else tb[i] = op3<kAndOr>(e[q], o[qn], tb[i + 1]);
        });
        sfor<n>([&](auto i_) __attribute__((always_inline)) {
            constexpr int i = decltype(i_)::value, q = line_sq(D, S0, i);
            constexpr bool hf = i >= 2, hb = i <= n - 3;
            if constexpr (MODE == 0) {
                if constexpr (hf && hb) L[q] = tf[i - 1] | tb[i + 1];
                else if constexpr (hf) L[q] = tf[i - 1];
                else L[q] = tb[i + 1];
            } else if constexpr (MODE == 1) {
                if constexpr (hf && hb) L[q] = op3<kOr3>(L[q], tf[i - 1], tb[i + 1]);
                else if constexpr (hf) L[q] |= tf[i - 1];
                else if constexpr (hb) L[q] |= tb[i + 1];
            } else {
                if constexpr (hf && hb) L[q] = op3<0xF8>(L[q], tf[i - 1] | tb[i + 1], mask);   // a | (b & c)
                else if constexpr (hf) L[q] = op3<0xF8>(L[q], tf[i - 1], mask);
                else if constexpr (hb) L[q] = op3<0xF8>(L[q], tb[i + 1], mask);
            }
        });
    }
}
template <int D, int MODE>
RAZ_SL void mobility_family(const uint32_t (&o)[64], const uint32_t (&e)[64], uint32_t (&L)[64], uint32_t mask) {
    sfor<64>([&](auto s_) __attribute__((always_inline)) {
        constexpr int s = decltype(s_)::value;
        if constexpr (line_start(D, s)) mobility_line<D, s, MODE>(o, e, L, mask);
    });
}
// L[s] = the boards on which the side `o` has a run ending on s; NOT yet masked with blank.  (The row family comes first: it writes
// every square of the board.)
RAZ_SL void mobility_raw(const uint32_t (&o)[64], const uint32_t (&e)[64], uint32_t (&L)[64]) {
    mobility_family<0, 0>(o, e, L, 0u);
    mobility_family<2, 1>(o, e, L, 0u);
    mobility_family<4, 1>(o, e, L, 0u);
    mobility_family<6, 1>(o, e, L, 0u);
}
// the same OR-ed into L on the boards of `mask` only
RAZ_SL void mobility_raw_into(const uint32_t (&o)[64], const uint32_t (&e)[64], uint32_t (&L)[64], uint32_t mask) {
    mobility_family<0, 2>(o, e, L, mask);
    mobility_family<2, 2>(o, e, L, mask);
    mobility_family<4, 2>(o, e, L, mask);
    mobility_family<6, 2>(o, e, L, mask);
}
// L[s] = the boards on which the side `o` may move to s
RAZ_SL void mobility(const uint32_t (&o)[64], const uint32_t (&e)[64], uint32_t (&L)[64]) {
    mobility_raw(o, e, L);
    sfor<64>([&](auto s_) __attribute__((always_inline)) {
        constexpr int s = decltype(s_)::value;
        L[s] = op3<kAndNor>(L[s], o[s], e[s]);
    });
}

// ---- calc_flip (lib/bitboard.py:70-92) for 32 boards, M[s] = the boards whose move is square s (at most one square per board).
// bb_flip_half's ray arithmetic along a line: from the move the run x[i] = e[q_i] & (M[q_(i-1)] | x[i-1]) extends over interior enemy
// squares; it is flipped where the first square behind it is own: c[i] = x[i] & (own[q_(i+1)] | c[i+1]); the same from the other end of
// the line; F |= c over the four families.  (The reference's rays are masked less than find_correct_moves' fills - the column ray runs
// over unmasked enemy discs, the diagonal rays over e_lr, lib/bitboard.py:84-92 - but a run that passes an END square of its line
// leaves the board behind it and closes nowhere: only interior squares can be flipped.)
// PRECONDITION: own & enemy == 0 on every board whose result is used.  On a board with a square of both colours the reference's
// `ray & (x + 1) & own` keeps every own-and-enemy square above the end of a run and `out - 1` spreads below the lowest of them: flips
// that are no lines of the board.  step_boards reports such boards (StepMasks.overlap) and the kernel steps them one by one.
constexpr bool flips_before(int d, int s) {   // an earlier family has written F[s]
    for (int f = 0; f < d; f += 2)
        if (passable(f, s)) return true;
    return false;
}
constexpr bool can_flip(int s) { return passable(0, s) || passable(2, s); }   // every square but the four corners
template <int D, int S0>
RAZ_SL void flips_line(const uint32_t (&o)[64], const uint32_t (&e)[64], const uint32_t (&rowsel)[8], const uint32_t (&colsel)[8], uint32_t (&F)[64]) {
    constexpr int n = line_len(D, S0);
    if constexpr (n >= 3) {
        uint32_t xf[8], xb[8], cf[8], cb[8], M[64];   // M: the line's squares only (M[s] = rowsel & colsel: made here, per family, instead of
        sfor<n>([&](auto i_) __attribute__((always_inline)) {   // being kept in 64 registers across all four)
            constexpr int q = line_sq(D, S0, decltype(i_)::value);
            M[q] = rowsel[q >> 3] & colsel[q & 7];
        });
        sfor<n - 2>([&](auto j_) __attribute__((always_inline)) {
            constexpr int i = 1 + decltype(j_)::value, q = line_sq(D, S0, i), qp = line_sq(D, S0, i - 1);
            if constexpr (i == 1) xf[i] = e[q] & M[qp];
            else xf[i] = op3<kAndOr>(e[q], M[qp], xf[i - 1]);
        });
        sfor<n - 2>([&](auto j_) __attribute__((always_inline)) {
            constexpr int i = n - 2 - decltype(j_)::value, q = line_sq(D, S0, i), qn = line_sq(D, S0, i + 1);
            if constexpr (i == n - 2) {
                xb[i] = e[q] & M[qn];
                cf[i] = xf[i] & o[qn];
            } else {
                xb[i] = op3<kAndOr>(e[q], M[qn], xb[i + 1]);
                cf[i] = op3<kAndOr>(xf[i], o[qn], cf[i + 1]);
            }
        });
        sfor<n - 2>([&](auto j_) __attribute__((always_inline)) {
            constexpr int i = 1 + decltype(j_)::value, q = line_sq(D, S0, i), qp = line_sq(D, S0, i - 1);
            if constexpr (i == 1) cb[i] = xb[i] & o[qp];
            else cb[i] = op3<kAndOr>(xb[i], o[qp], cb[i - 1]);
            if constexpr (flips_before(D, q)) F[q] = op3<kOr3>(F[q], cf[i], cb[i]);
            else F[q] = cf[i] | cb[i];
        });
    }
}
template <int D>
RAZ_SL void flips_family(const uint32_t (&o)[64], const uint32_t (&e)[64], const uint32_t (&rowsel)[8], const uint32_t (&colsel)[8], uint32_t (&F)[64]) {
    sfor<64>([&](auto s_) __attribute__((always_inline)) {
        constexpr int s = decltype(s_)::value;
        if constexpr (line_start(D, s)) flips_line<D, s>(o, e, rowsel, colsel, F);
    });
}
// F[s] for every square but the corners (which no move flips: F[0], F[7], F[56], F[63] stay unwritten).  The boards that play on square
// s are rowsel[s >> 3] & colsel[s & 7].
RAZ_SL void flips(const uint32_t (&o)[64], const uint32_t (&e)[64], const uint32_t (&rowsel)[8], const uint32_t (&colsel)[8], uint32_t (&F)[64]) {
    flips_family<0>(o, e, rowsel, colsel, F);
    flips_family<2>(o, e, rowsel, colsel, F);
    flips_family<4>(o, e, rowsel, colsel, F);
    flips_family<6>(o, e, rowsel, colsel, F);
}

// rows of a superblock: lane l's pair of row r is element (sb * 1024 + r * 64 + l) of the ulonglong2 view
RAZ_SL void load_boards(const ulonglong2* __restrict__ src, size_t at, uint32_t (&a)[64]) {
    sfor<16>([&](auto r_) __attribute__((always_inline)) {
        constexpr int r = decltype(r_)::value;
        const ulonglong2 v = src[at + (size_t)r * 64];
        a[2 * r] = (uint32_t)v.x;
        a[32 + 2 * r] = (uint32_t)(v.x >> 32);
        a[2 * r + 1] = (uint32_t)v.y;
        a[32 + 2 * r + 1] = (uint32_t)(v.y >> 32);
    });
}
RAZ_SL void store_boards(ulonglong2* __restrict__ dst, size_t at, const uint32_t (&a)[64]) {
    sfor<16>([&](auto r_) __attribute__((always_inline)) {
        constexpr int r = decltype(r_)::value;
        ulonglong2 v;
        v.x = ((unsigned long long)a[32 + 2 * r] << 32) | a[2 * r];
        v.y = ((unsigned long long)a[32 + 2 * r + 1] << 32) | a[2 * r + 1];
        dst[at + (size_t)r * 64] = v;
    });
}

// ---- bytes -> planes.  w[q] holds one byte for each of the boards 4q .. 4q + 3 (byte i = board 4q + i); plane[b] receives bit b of
// the 32 bytes, one board per bit: 8 x 8 bit transposes of the bytes of 8 boards (bb_flip_diag_a1h8: byte f, bit r <-> byte r, bit f),
// then a 4 x 4 byte transpose across the four groups.
RAZ_SL void planes_of_bytes(const uint32_t (&w)[8], uint32_t (&plane)[8]) {
    uint32_t lo[4], hi[4];
    sfor<4>([&](auto g_) __attribute__((always_inline)) {
        constexpr int g = decltype(g_)::value;
        const raz_bb t = bb_flip_diag_a1h8(((raz_bb)w[2 * g + 1] << 32) | w[2 * g]);   // byte b = bit b of the boards 8g .. 8g + 7
        lo[g] = (uint32_t)t;
        hi[g] = (uint32_t)(t >> 32);
    });
    // plane[b] = {lo[0].b, lo[1].b, lo[2].b, lo[3].b} (b < 4), the same of hi for b >= 4
    const uint32_t l01a = perm<0x05010400u>(lo[1], lo[0]), l01b = perm<0x07030602u>(lo[1], lo[0]);   // {0.b0,1.b0,0.b1,1.b1}, {0.b2,1.b2,0.b3,1.b3}
    const uint32_t l23a = perm<0x05010400u>(lo[3], lo[2]), l23b = perm<0x07030602u>(lo[3], lo[2]);
    plane[0] = perm<0x05040100u>(l23a, l01a);
    plane[1] = perm<0x07060302u>(l23a, l01a);
    plane[2] = perm<0x05040100u>(l23b, l01b);
    plane[3] = perm<0x07060302u>(l23b, l01b);
    const uint32_t h01a = perm<0x05010400u>(hi[1], hi[0]), h01b = perm<0x07030602u>(hi[1], hi[0]);
    const uint32_t h23a = perm<0x05010400u>(hi[3], hi[2]), h23b = perm<0x07030602u>(hi[3], hi[2]);
    plane[4] = perm<0x05040100u>(h23a, h01a);
    plane[5] = perm<0x07060302u>(h23a, h01a);
    plane[6] = perm<0x05040100u>(h23b, h01b);
    plane[7] = perm<0x07060302u>(h23b, h01b);
}

// most significant bit of every byte of x that is non-zero (SWAR)
RAZ_SL uint32_t nonzero_bytes(uint32_t x) { return (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u; }

// OR of the 64 words
RAZ_SL uint32_t any_square(const uint32_t (&a)[64]) {
    uint32_t r[22];
    sfor<21>([&](auto i_) __attribute__((always_inline)) {
        constexpr int i = decltype(i_)::value;
        r[i] = op3<kOr3>(a[3 * i], a[3 * i + 1], a[3 * i + 2]);
    });
    r[21] = a[63];
    uint32_t q[8];
    sfor<7>([&](auto i_) __attribute__((always_inline)) {
        constexpr int i = decltype(i_)::value;
        q[i] = op3<kOr3>(r[3 * i], r[3 * i + 1], r[3 * i + 2]);
    });
    q[7] = r[21];
    return op3<kOr3>(op3<kOr3>(q[0], q[1], q[2]), op3<kOr3>(q[3], q[4], q[5]), q[6] | q[7]);
}

// ---- ReversiEnv.step (env/reversi_env.py:42-85, bb_env_step) for the 32 boards of a lane.
//   b / w: black / white of the boards, SLICED (in: before the step, out: after it);  L (out): the legal moves of the side to move
//   after the step, sliced;  pw / sw / aw: player / status / action bytes, 4 boards per word (byte i of word q = board 4q + i);
//   pw / sw are updated.  nb_nw(k, nb, nw) must return the disc counts of board k AFTER the step (the caller has the boards in memory
//   layout when it is asked: it is called after `finish` below).
// What a board does: finished (status != 0): nothing, legal = 0 (the sweep's contract, raz_sweep.hip step_first).  action 255: the
// other side wins | RESIGNED.  No disc flipped (also: an action outside 0..63): the other side wins | ILLEGAL, board untouched.
// Else the move is made; the opponent moves next if it can, else the mover again if it can, else the discs are counted.
struct StepMasks {
    uint32_t moved, nz1, nz2;   // one bit per board: the move flipped something / the opponent can move / (else) the mover can
    uint32_t overlap;           // the board came in with a square that is black AND white: see below
};
// Boards on which black & white != 0 are not positions of the game (ReversiEnv never produces one from the opening; a move onto an
// occupied enemy square does).  bb_calc_flip's ray arithmetic gives such a board flips that are no lines of the board - `ray & (x + 1) &
// own` keeps every own-and-enemy square above the end of a run, and `out - 1` then spreads below the lowest of them - which the
// square-by-square walks here do not reproduce.  step_boards reports these boards; the kernel steps a lane that holds one board by
// board with the reference-shaped arithmetic instead (k_step_sliced), so the results are the reference's for EVERY input.
RAZ_SL StepMasks step_boards(uint32_t (&b)[64], uint32_t (&w)[64], uint32_t (&L)[64], const uint32_t (&pw)[8], const uint32_t (&sw)[8],
                             const uint32_t (&aw)[8]) {
    // one code byte per board: bits 0-5 the action's square, bit 6 "plays a move" (running game, action 0..63), bit 7 "black to move"
    uint32_t code[8], plane[8];
    sfor<8>([&](auto q_) __attribute__((always_inline)) {
        constexpr int q = decltype(q_)::value;
        const uint32_t a = aw[q];
        const uint32_t idle = nonzero_bytes(sw[q]) | ((a | (a << 1)) & 0x80808080u);   // finished, or action >= 64 (255 = resign included)
        const uint32_t white = nonzero_bytes(pw[q] ^ 0x01010101u);                      // player != 1 (bb_env_step: anything but black is white)
        code[q] = (a & 0x3f3f3f3fu) | ((~idle & 0x80808080u) >> 1) | (~white & 0x80808080u);
    });
    planes_of_bytes(code, plane);
    const uint32_t pb = plane[7], plays = plane[6];
    // own / enemy by the side to move, in place: b = own, w = enemy
    uint32_t both = 0;
    sfor<64>([&](auto s_) __attribute__((always_inline)) {
        constexpr int s = decltype(s_)::value;
        const uint32_t bs = b[s], ws = w[s];
        both = op3<0xEA>(bs, ws, both);
        b[s] = op3<kSelect>(pb, bs, ws);
        w[s] = op3<kSelect>(pb, ws, bs);
    });
    // the boards that play on square s: rowsel[s >> 3] & colsel[s & 7]
    uint32_t rowsel[8], colsel[8], F[64];
    sfor<8>([&](auto i_) __attribute__((always_inline)) {
        constexpr int i = decltype(i_)::value;
        colsel[i] = op3<(1 << i)>(plane[2], plane[1], plane[0]);
        rowsel[i] = op3<(1 << i)>(plane[5], plane[4], plane[3]) & plays;
    });
    RAZ_SL_PHASE();
    flips(b, w, rowsel, colsel, F);
    RAZ_SL_PHASE();
    F[0] = F[7] = F[56] = F[63] = 0;   // (the corners: never flipped)
    StepMasks m;
    m.overlap = both;
    m.moved = any_square(F);
    sfor<64>([&](auto s_) __attribute__((always_inline)) {   // own ^= flipped; own |= 1 << action; enemy ^= flipped - only where something flipped
        constexpr int s = decltype(s_)::value;
        const uint32_t mv = op3<0x80>(rowsel[s >> 3], colsel[s & 7], m.moved);
        if constexpr (can_flip(s)) {
            b[s] = op3<0xBE>(b[s], F[s], mv);   // (own ^ flipped) | move: XOR as the reference has it
            w[s] = w[s] ^ F[s];
        } else {
            b[s] |= mv;
        }
    });
    // the opponent's moves (:66) ...
    RAZ_SL_PHASE();
    mobility_raw(w, b, L);
    sfor<64>([&](auto s_) __attribute__((always_inline)) {
        constexpr int s = decltype(s_)::value;
        L[s] = op3<kAndNor>(L[s], b[s], w[s]);
    });
    m.nz1 = any_square(L) & m.moved;
    // ... and where it has none the mover's own (:68), OR-ed in on those boards only (their L is empty so far)
    const uint32_t second = m.moved & ~m.nz1;
    RAZ_SL_PHASE();
    mobility_raw_into(b, w, L, second);
    RAZ_SL_PHASE();
    sfor<64>([&](auto s_) __attribute__((always_inline)) {
        constexpr int s = decltype(s_)::value;
        L[s] = op3<kAndNor>(L[s] & m.moved, b[s], w[s]);   // blank squares, of boards that moved (blank twice is blank)
        const uint32_t os = b[s], es = w[s];                // back to black / white
        b[s] = op3<kSelect>(pb, os, es);
        w[s] = op3<kSelect>(pb, es, os);
    });
    m.nz2 = any_square(L);   // (read where nz1 is clear only)
    return m;
}

// ---- the second half of a step on its own, for a kernel that has made the moves board by board (k_step_hybrid): o / e = the mover's and
// the opponent's discs AFTER the move, sliced; moved = the boards that moved.  L: the legal moves of whoever moves next on those boards
// (the opponent's, :66; where it has none the mover's own, :68), nothing on the others.  nz1 / nz2 as StepMasks.
RAZ_SL void legal_after_move(const uint32_t (&o)[64], const uint32_t (&e)[64], uint32_t (&L)[64], uint32_t moved, uint32_t& nz1, uint32_t& nz2) {
    mobility_raw(e, o, L);
    sfor<64>([&](auto s_) __attribute__((always_inline)) {
        constexpr int s = decltype(s_)::value;
        L[s] = op3<kAndNor>(L[s], o[s], e[s]);
    });
    nz1 = any_square(L) & moved;
    const uint32_t second = moved & ~nz1;
    RAZ_SL_PHASE();
    mobility_raw_into(o, e, L, second);
    RAZ_SL_PHASE();
    sfor<64>([&](auto s_) __attribute__((always_inline)) {
        constexpr int s = decltype(s_)::value;
        L[s] = op3<kAndNor>(L[s] & moved, o[s], e[s]);
    });
    nz2 = any_square(L);
}

}  // namespace sl
