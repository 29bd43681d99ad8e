/* raz.h — C ABI of libraz, the MI355X-native self-play hot path that sits behind the
 * ReversiEnv / ReversiPlayer / play_*.json surface of mokemokechicken/reversi-alpha-zero.
 *
 * The reference has no FFI of its own: its seams are Python call signatures, and its only native
 * code is the Cython pair lib/alt/bitboard_cython.pyx + lib/alt/reversi_solver_cython.pyx
 * (`cpdef unsigned long long f(unsigned long long, unsigned long long)`).  Each entry point below
 * names the reference interface it replaces (file:line under /root/reference/src/reversi_zero).
 *
 * Conventions
 *   - plain C types only; device pointers are ordinary pointers into HBM owned by the CALLER
 *     (libraz never allocates or frees device memory);
 *   - `raz_stream_t` is a hipStream_t passed as void* (NULL = the null stream); batched calls are
 *     asynchronous on that stream;
 *   - functions returning int return 0 on success and a negative RAZ_E* code on failure, with a
 *     message retrievable through raz_last_error() (thread-local); nothing aborts the process;
 *   - bitboards: bit i = square i, bit 0 = top-left, bit 63 = bottom-right (lib/bitboard.py:10-17);
 *     player 1 = black, 2 = white (env/reversi_env.py:9); winner 1/2/3 = black/white/draw (:11);
 *   - an engine handle is NOT re-entrant: one host thread drives one handle on one device.
 */
#ifndef RAZ_H
#define RAZ_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAZ_ABI_VERSION 1

#define RAZ_OK 0
#define RAZ_EINVAL (-1)   /* bad argument (range, NULL, alignment)            */
#define RAZ_EDEVICE (-2)  /* HIP runtime error (no device, launch failure)    */
#define RAZ_ENOMEM (-3)   /* caller-provided workspace too small              */
#define RAZ_ESTATE (-4)   /* call not valid in the handle's current state     */

typedef void* raz_stream_t;

int raz_abi_version(void);
const char* raz_last_error(void);

/* ---- scalar host primitives: lib/bitboard.py (precedent: lib/alt/bitboard_cython.pyx) -------- */
uint64_t raz_find_correct_moves(uint64_t own, uint64_t enemy);  /* bitboard.py:53  / pyx:1   */
/* bitboard.py:70 / pyx:18.  pos must be 0..63 (the reference asserts, :78): outside that range
 * the call returns 0 and records RAZ_EINVAL in raz_last_error(); the Python wrapper asserts. */
uint64_t raz_calc_flip(int pos, uint64_t own, uint64_t enemy);
int raz_bit_count(uint64_t x);                                   /* bitboard.py:132 / pyx:97  */
uint64_t raz_flip_vertical(uint64_t x);                          /* bitboard.py:119 / pyx:43  */
uint64_t raz_flip_diag_a1h8(uint64_t x);                         /* bitboard.py:141 / pyx:52  */
uint64_t raz_rotate90(uint64_t x);                               /* bitboard.py:154 / pyx:64  */
uint64_t raz_rotate180(uint64_t x);                              /* bitboard.py:158 / pyx:68  */
/* bit_to_array(x, size) (bitboard.py:136): out[i] = bit i of x, i < size <= 64. */
int raz_bit_to_array(uint64_t x, int size, uint8_t* out);

/* ReversiEnv.step (env/reversi_env.py:42-74) on one position held by the caller.
 * action 0..63, or 255 for the reference's `None` (resign).  On return *status is 0 (running) or
 * winner 1/2/3 | 0x10 (ended by an illegal no-flip move) | 0x20 (ended by resignation);
 * *legal = legal-move mask of the side to move (0 when done). */
int raz_env_step(uint64_t* black, uint64_t* white, uint8_t* player, uint8_t* status,
                 uint64_t* legal, int action);

/* ---- batched device sweeps: one board per lane over SoA arrays in HBM ----------------------- */
/* find_correct_moves over n boards.  24 B/board. */
int raz_legal_moves_batch(const uint64_t* own, const uint64_t* enemy, uint64_t* legal, size_t n,
                          raz_stream_t stream);
/* calc_flip over n (pos, own, enemy) triples; pos[i] > 63 yields 0.  25 B/board. */
int raz_calc_flip_batch(const uint8_t* pos, const uint64_t* own, const uint64_t* enemy,
                        uint64_t* flipped, size_t n, raz_stream_t stream);
/* ReversiEnv.step over n games, in place.  Games whose status[i] != 0 on entry are finished and
 * are left untouched with legal[i] = 0 (the reference's loops never step a done env,
 * worker/self_play.py:155).  action[i] 0..63 or 255 (resign).  45 B/board:
 * reads black,white (16) player,status,action (3); writes black,white (16) player,status (2) legal (8). */
int raz_step_batch(uint64_t* black, uint64_t* white, uint8_t* player, uint8_t* status,
                   uint64_t* legal, const uint8_t* action, size_t n, raz_stream_t stream);
/* _game_over scoring (env/reversi_env.py:76-85): winner[i] in {1,2,3} by disc count,
 * diff[i] = popcount(black) - popcount(white).  18 B/board. */
int raz_score_batch(const uint64_t* black, const uint64_t* white, uint8_t* winner, int8_t* diff,
                    size_t n, raz_stream_t stream);
/* Dihedral transform used before the net and for saved rows (agent/player.py:300-305,169-178):
 * sym[i] = flip*4 + rot, flip_vertical first, then rot right-rotations.  17 B/board. */
int raz_d4_batch(const uint64_t* in, uint64_t* out, const uint8_t* sym, size_t n,
                 raz_stream_t stream);
/* bit_to_array for the net input (agent/player.py:307-309, worker/optimize.py:225):
 * planes[i][0][sq] = bit sq of own[i], planes[i][1][sq] = bit sq of enemy[i], as float32. */
int raz_planes_batch(const uint64_t* own, const uint64_t* enemy, float* planes, size_t n,
                     raz_stream_t stream);
/* Uniform random playout driver for tests/benches (test infrastructure shipped with the library,
 * mirrors SURVEY §8(c) "random playout"): action[i] = the k-th set bit of legal[i] where
 * k = rnd[i] % popcount(legal[i]); 255 when legal[i] == 0. */
int raz_pick_kth_legal_batch(const uint64_t* legal, const uint32_t* rnd, uint8_t* action, size_t n,
                             raz_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RAZ_H */
