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

#define RAZ_ABI_VERSION 3   /* 2: compact tree nodes - raz_engine_config.pool_bytes_per_game, raz_engine_stats.max_pool_bytes.
                               3: the end-game solver is a pool of worker waves shared by all games - raz_engine_config.solver_pool_waves,
                                  reserved bits 16-23 now budget the pool's launches; raz_engine_uses_graph is gone, raz_net_range_stats
                                  was added, raz_engine_config.reserved bit 2 and raz_net.reserved 5/6 are rejected */

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
/* Which KERNEL FORM the whole superblocks (2048 boards) of a batch of n boards run on - large batches take the bit-sliced kernels
 * (32 boards per lane: csrc/raz_sweep_sliced.h), the rest of the batch and small batches the board-per-lane ones; results are identical.
 * The thresholds are read once per process (RAZ_SWEEP_SLICED_MIN=<boards> overrides them; RAZ_SWEEP_SLICED_STEP=0/1/2 forces the step form).
 *   *legal_moves_form: 0 = k_legal_moves (a board per lane), 1 = k_legal_moves_sliced (from 2^25 boards on);
 *   *step_form:        0 = k_step (a board per lane), 1 = k_step_sliced, 2 = k_step_hybrid (from 2^26 boards on; needs the player /
 *                      status / action arrays 16-byte aligned - a batch whose arrays are not stays on k_step).
 * For tests and benches that must know what they measured; no device work. */
int raz_sweep_forms(size_t n, int* legal_moves_form, int* step_form);
/* Uniform random playout driver for tests/benches (test infrastructure shipped with the library,
 * mirrors SURVEY §8(c) "random playout"): action[i] = the k-th set bit of legal[i] where
 * k = rnd[i] % popcount(legal[i]); 255 when legal[i] == 0. */
int raz_pick_kth_legal_batch(const uint64_t* legal, const uint32_t* rnd, uint8_t* action, size_t n,
                             raz_stream_t stream);

/* ---- policy/value net: agent/model.py:28-72 evaluated as agent/api.py:30-45 predict() ------------
 * Weights arrive as the host-side "raznet v1" blob (int32[8] header {0x4E5A4152,1,F,R,V,3,0,0} +
 * float32 conv/dense parameters with BatchNorm folded; written by
 * reversi_alpha_zero_amd.agent.model.ReversiNet.to_blob).  The caller owns the device buffers. */
typedef struct {
    int32_t filters, res_layers, value_fc, reserved;
    void* d_weights;      /* device, >= raz_net_weight_bytes() */
    size_t weight_bytes;
} raz_net;

size_t raz_net_weight_bytes(int filters, int res_layers, int value_fc); /* 0 = unsupported shape */
size_t raz_net_scratch_bytes(int filters, int value_fc, size_t n);      /* HBM scratch for n positions */
int raz_net_load(raz_net* net, const void* blob, size_t blob_bytes, void* d_weights, size_t d_bytes,
                 raz_stream_t stream);
/* ReversiModelAPI.predict (agent/api.py:30) for n positions given as (own, enemy) bitboards of the
 * side to move — the planes of agent/player.py:307-309 are formed on the fly.  policy: n x 64
 * float32 (softmax), value: n float32 (tanh).  active (nullable): positions with active[i]==0 are
 * skipped and their outputs left untouched. */
int raz_net_forward(const raz_net* net, const uint64_t* own, const uint64_t* enemy,
                    const uint8_t* active, float* policy, float* value, size_t n, void* scratch,
                    size_t scratch_bytes, raz_stream_t stream);

/* raz_net.reserved selects the forward kernels: 0 = the exact-f32 kernels chosen by shape ("raznet-forward-v1": every output
 * one k-ordered fmaf chain, bit-identical to the CPU oracle); 4 (filters % 128 == 0) = "raznet-forward-v2": the 3x3 trunk on
 * the f16 matrix cores with every f32 operand split into two halfs (csrc/raz_net_f16x3.hip: 3 f16 MFMAs per product, f32
 * accumulation, within 1e-5 of the fp32 graph, 16/3 of the f32-MFMA rate); 1, 2: test variants of v1 (same bits).  v2's split activations must
 * stay inside the f16 range: a row (position) whose activations do not is evaluated by the exact-f32 chains inside the same
 * forward; *overflowed = 1 reports that some forward since raz_net_load had more such rows than it repairs (32; sticky): run the
 * net with reserved = 0 then.  Synchronises `stream`. */
int raz_net_range_check(const raz_net* net, int* overflowed, raz_stream_t stream);
/* The same plus *rows_repaired: v2 rows (positions) since raz_net_load whose activations left the f16 range and were therefore
 * evaluated by the exact-f32 chains instead, inside the forward that met them (at most 32 rows per forward; a forward with more
 * raises the sticky flag reported by *overflowed).  A row's answer is a function of its position alone either way.  No reference
 * counterpart (Keras computes in fp32 throughout, agent/api.py:30-45).  Synchronises `stream`. */
int raz_net_range_stats(const raz_net* net, int* overflowed, unsigned long long* rows_repaired, raz_stream_t stream);

/* ---- batched self-play engine --------------------------------------------------------------------
 * Replaces, for n_games concurrent games, SelfPlayWorker.start_game (worker/self_play.py:139-175)
 * driving two ReversiPlayer objects (agent/player.py:28-428) through ReversiEnv, with the leaf
 * evaluations of ALL live games gathered into one raz_net_forward batch per simulation step
 * (the reference batches at most prediction_queue_size=16 leaves per worker, player.py:329-355,
 * plus Pipe fan-in, api.py:75-100).  parallel_search_num simulations are in flight per game: 1 is the
 * reference's reproducible mode (SURVEY.md §7 hard part 1); 2..16 follow the reference's asyncio event
 * loop in exact virtual time (raz-sched-v1), each in-flight leaf being one row of the batch.
 *
 * Field names follow PlayConfig (config.py:128-166).  Randomness: raz-rng-v1 keyed by
 * (seed, global game id), so results do not depend on n_games, slot order or GPU count. */
typedef struct {
    int32_t thinking_loop;                    /* config.py:133 */
    int32_t required_visit_to_decide_action;  /* :134 */
    int32_t start_rethinking_turn;            /* :135 */
    int32_t change_tau_turn;                  /* :139 */
    int32_t virtual_loss;                     /* :140 (with parallel_search_num=1 only its rounding is observable) */
    int32_t allowed_resign_turn;              /* :146 */
    int32_t has_resign_threshold;             /* resign_threshold is not None */
    int32_t share_mtcs_info;                  /* share_mtcs_info_in_self_play :131 */
    int32_t mirror_updates;                   /* 1: also maintain the colour-mirrored key like player.py:279-280,
                                                 323-324 (required when share_mtcs_info=1); 0: skip those writes
                                                 (dead unless a colour-swapped transposition occurs) */
    int32_t record_root_w;                    /* 1: also record root W per ply (parity tests) */
    double c_puct;                            /* :136 */
    double noise_eps;                         /* :137 */
    double dirichlet_alpha;                   /* :138, any alpha > 0 (0.5: Box-Muller pairs; < 1 / > 1: numpy's legacy gamma schemes) */
    double resign_threshold;                  /* :145 */
    double disable_resignation_rate;          /* :147 */
    uint32_t n_games;                         /* game slots in flight (B) */
    uint32_t nodes_per_game;                  /* most tree nodes a game's pool may hold (sizes the hash table and the node directory) */
    uint32_t table_slots;                     /* hash slots per game, power of two >= 2*nodes_per_game */
    uint32_t max_plies;                       /* record capacity per game (>= 64) */
    uint32_t seed;
    uint32_t reserved;                        /* bit 0: in-kernel phase profile; bit 1: single stream; bit 3: drive even
                                                 parallel_search_num <= 1 with the slot kernel (tests); bit 4: 16-filter nets: tree and net in
                                                 ONE kernel, the game's wave evaluates its own leaves (csrc/raz_engine_fused.hip
                                                 k_tree_net / k_tree_par_net; same results); bits 8-11:
                                                 slices/streams (0 = 3); bits 12-15: max simulations per game per tree launch
                                                 (0 = 2; slot kernel: simulations STARTED per launch beyond parallel_search_num);
                                                 bits 16-23: x 64 = iterations a worker lane of the end-game solver's pool runs between two
                                                 tree launches (0 = the default, 96); a game whose solve - the root's or one inside a simulation - is
                                                 not answered yet stays suspended: results do not depend on the value;
                                                 bits 24-27: tree launches per round of that pool (0 = the library's default; 1 = every
                                                 step waits for the pool's round; n > 1 = the round runs on a stream of its own beside
                                                 the next n - 1 steps, so that games that wait for no solve are not held up by it;
                                                 raz_engine_step joins it before it returns).  Results do not depend on it either.
                                                 Every other bit must be 0 (RAZ_EINVAL) */
    int32_t use_solver_turn;                  /* config.py:154: 0 = off, else >= 46: exact end-game solve at the root
                                                 (agent/player.py:100-103,150-161; lib/alt/reversi_solver_cython.pyx) */
    int32_t use_solver_turn_in_simulation;    /* config.py:155: 0 = off, else >= 46: win/loss solve inside simulations
                                                 (agent/player.py:237-251) */
    uint32_t solver_memo_slots;               /* per-game memo of solved positions, power of two (0 with the solver off) */
    uint32_t parallel_search_num;             /* config.py:142: simulations in flight per game; 0/1 = one (the reference's
                                                 reproducible mode), 2..16 = the asyncio loop in exact virtual time
                                                 (raz-sched-v1), bit-exact vs the reference on such a loop */
    uint64_t pool_bytes_per_game;             /* bytes of a game's node pool.  Nodes are compact and variable-size: 40 B + 20 B per
                                                 LEGAL move of the position (the reference keeps three f64[64] per key = 1536 B,
                                                 agent/player.py:62-66), ~212 B on average over a game, 704 B at most.
                                                 0 = nodes_per_game x 232 + 64 x 704; at most 256 MB */
    uint32_t solver_pool_waves;               /* worker wavefronts of the end-game solver's pool (csrc/raz_solver_pool.h): positions of 5..14
                                                 empties are solved by a pool of lanes shared by all games, one subtree per lane, instead
                                                 of inside the game's own wave.  0 = one per four games, at most 1280 (what the chip's LDS holds at once); 36 KB
                                                 of workspace each.  Results do not depend on the value.  No reference counterpart
                                                 (lib/alt/reversi_solver_cython.pyx runs one position at a time) */
    uint32_t reserved2;                       /* must be 0 */
} raz_engine_config;

typedef struct raz_engine raz_engine; /* opaque host handle; not re-entrant */

typedef struct {
    uint64_t finished_games;  /* games whose status != 0 */
    uint64_t total_sims;      /* start_search_my_move invocations executed (SURVEY §8(d) metric) */
    uint64_t nn_leaves;       /* leaf positions sent to the net */
    uint64_t error_flags;     /* 0 = ok; 1 node pool full, 2 table full, 4 records full, 8 path overflow */
    uint64_t selections;      /* select_action_q_and_u calls (sum of descent depths) */
    uint64_t max_pool_used;   /* most nodes in a running game's pool (limit: nodes_per_game) */
    uint64_t idle_or_done;    /* slots that are idle (never started / one-move mode finished) or finished */
    uint64_t max_pool_bytes;  /* most bytes used in a running game's pool (limit: pool_bytes_per_game) */
} raz_engine_stats;

size_t raz_engine_workspace_bytes(const raz_engine_config* cfg);
/* d_workspace: device buffer of at least raz_engine_workspace_bytes(), 256-byte aligned, owned by
 * the caller for the lifetime of the handle.  net: loaded with raz_net_load. */
int raz_engine_create(const raz_engine_config* cfg, const raz_net* net, void* d_workspace,
                      size_t workspace_bytes, void* d_net_scratch, size_t net_scratch_bytes,
                      raz_engine** out);
void raz_engine_destroy(raz_engine* e);
/* Reset every slot to a fresh game: slot i plays global game id first_game_id + i with
 * sims_per_move[i] simulations per move (host array of n_games entries; the reference decides this
 * per game from the schedule, worker/self_play.py:145,262-272).  Slots i >= n_active stay idle. */
int raz_engine_start(raz_engine* e, uint32_t first_game_id, const uint32_t* sims_per_move,
                     uint32_t n_active, raz_stream_t stream);
/* The next game of every slot ON THE SLOT'S TREE: SelfPlayWorker.start keeps its MCTSInfo for
 * reset_mtcs_info_per_game games (worker/self_play.py:109-111,132-134; mini.yml ships 3).  Boards, records,
 * random-stream counters and statistics start afresh for global game ids first_game_id + i; nodes, tables and
 * pools stay, and every key that holds a prior counts as expanded for the new game's players
 * (agent/player.py:47).  All games of the previous round must have finished and their records been read.
 * Without share_mtcs_info the reference carries nothing over and this is raz_engine_start. */
int raz_engine_next_game(raz_engine* e, uint32_t first_game_id, const uint32_t* sims_per_move,
                         uint32_t n_active, raz_stream_t stream);
/* Enqueue n_steps simulation steps (each: tree kernel = backup + move logic + select, then one
 * net batch over the gathered leaves).  Asynchronous w.r.t. the host; all work is ordered after
 * prior work on `stream` and before later work on it.  With n_games >= 256 the batch is stepped as
 * 3 slices on `stream` and two internal streams so that one slice's net kernel overlaps the
 * other slices' tree kernels. */
int raz_engine_step(raz_engine* e, uint32_t n_steps, raz_stream_t stream);
/* Same as raz_engine_step, with HIP events recorded around every kernel launch on the stream it runs
 * on: the summed durations (ms) of the tree-kernel launches and of the net-kernel launches are
 * ADDED to *tree_ms / *net_ms (with n_games >= 256 a step is 2 launches of each kernel, one per
 * half batch, on two streams).  Synchronises the stream. */
int raz_engine_step_timed(raz_engine* e, uint32_t n_steps, double* tree_ms, double* net_ms,
                          raz_stream_t stream);
/* ReversiPlayer.action_with_evaluation(own, enemy) (agent/player.py:82-134) for the player that owns
 * slot `slot`: put the slot on (black, white, player to move) keeping its tree (the reference keeps
 * var_n/var_w/var_p across calls), its random-stream counters and its records, and arm one move of
 * `sims` simulations.  one_move != 0: the slot idles once the move is decided (read it back with
 * raz_engine_read_records: the last recorded ply); 0: the game simply continues from that position. */
int raz_engine_set_position(raz_engine* e, uint32_t slot, uint64_t black, uint64_t white, int player,
                            uint32_t sims, int enable_resign, int one_move, raz_stream_t stream);
/* raz_engine_set_position for slots first_slot .. first_slot + n - 1 in one launch: (black, white, player to move) of slot
 * first_slot + i in DEVICE arrays d_black[i], d_white[i], d_player[i] (players 1 / 2; the caller guarantees the positions are
 * playable, as with the scalar call), the same sims / enable_resign / one_move for all.  Asynchronous on `stream`. */
int raz_engine_set_positions(raz_engine* e, uint32_t first_slot, uint32_t n, const uint64_t* d_black, const uint64_t* d_white,
                             const uint8_t* d_player, uint32_t sims, int enable_resign, int one_move, raz_stream_t stream);
/* ReversiPlayer.stop_thinking (agent/player.py:163-164): the slot's running search ends at the next step
 * and the move is decided from the tree as it is. */
int raz_engine_stop_thinking(raz_engine* e, uint32_t slot, raz_stream_t stream);
/* A ReversiPlayer constructed on a used MCTSInfo takes expanded = set(var_p.keys()) (agent/player.py:47):
 * mark every key of the slot that holds a prior as expanded for player index 0/1. */
int raz_engine_adopt_tree(raz_engine* e, uint32_t slot, int player_index, raz_stream_t stream);
/* MCTSInfo introspection (agent/player.py:22,63-69: var_n[key], var_w[key], var_p[key] with
 * key = CounterKey(black, white, next_player)): copies slot `slot`'s statistics of that key into host
 * arrays of 64 (any may be NULL).  owner: 0 when share_mtcs_info, else the player index (0 black,
 * 1 white) whose tree is read.  *found = 0 and zeros (the defaultdict default) when the key was never
 * touched; the tree is not modified.  p64 is the masked, normalised prior the search uses
 * (var_p after expand_and_evaluate, player.py:283-327).  Synchronises `stream`. */
int raz_engine_read_node(raz_engine* e, uint32_t slot, uint64_t black, uint64_t white, int next_player,
                         int owner, double* w64, uint32_t* n64, float* p64, int* found, raz_stream_t stream);
/* Prune the nodes no future search can reach (positions with fewer discs than the current real
 * position; the disc count only grows) in every game whose pool holds >= threshold nodes, compacting
 * the pool and rebuilding that game's table.  Does not change any result.  Asynchronous.  A pool is full when EITHER its node
 * count reaches nodes_per_game or its bytes reach pool_bytes_per_game (error flag 1): prune when
 * max_pool_used / max_pool_bytes of raz_engine_stats_sync come close (a simulation adds at most two nodes of <= 704 B). */
int raz_engine_gc(raz_engine* e, uint32_t threshold, raz_stream_t stream);
/* Number of slices/streams a step is split into (1..8; 1 = one tree launch + one net launch over the
 * whole batch).  Call with the stream idle. */
int raz_engine_set_parts(raz_engine* e, int parts);
/* Tree launches per round of the end-game solver's pool (raz_engine_config.reserved bits 24-27 set it at creation): 0 = the library's
 * default, 1 = every step waits for the pool's round, n in 2..15 = the round runs on a stream of its own beside the next n - 1 steps.
 * A batch whose games all reach the solver together (lock-step whole games) is served best by 1, continuous batching - a sixth of
 * the slots in the end game at any time - by 3 (mini.yml as shipped, 4096 slots: 22.7 M -> 30.3 M sims/s).  No result depends on it.
 * Call between raz_engine_step calls. */
int raz_engine_set_solver_pool_every(raz_engine* e, int n);
/* Synchronise the stream and read the counters. */
int raz_engine_stats_sync(raz_engine* e, raz_engine_stats* out, raz_stream_t stream);
/* Copy finished-game records to host memory (synchronous).  headers: n_games*max_plies*48 bytes
 * (struct layout in csrc/raz_engine.h: own u64, enemy u64, n f64, q f64, action i8, player u8,
 * turn u8, has_row u8, sims u32, loops u32, pad u32); root_n: n_games*max_plies*64 u32;
 * root_w: same count of f64 or NULL; n_plies, status (winner|flags), resigned[2], game_id,
 * enable_resign: per game.  Any pointer may be NULL to skip that array. */
int raz_engine_read_records(raz_engine* e, void* headers, uint32_t* root_n, double* root_w,
                            uint32_t* n_plies, uint8_t* status, uint8_t* resigned,
                            uint32_t* game_id, uint8_t* enable_resign, uint64_t* final_black,
                            uint64_t* final_white, raz_stream_t stream);
/* The unit of the record gather ("gather of finished-game records to rank 0"): the records of slots
 * [first_slot, first_slot + n_slots), cut to `plies` plies per game (raz_engine_records_extent gives the largest
 * n_plies; unused plies are zero), packed device-to-device into dense caller arrays that a collective can move as
 * they are: d_headers n_slots*plies*48 bytes, d_root_n n_slots*plies*64 u32, d_summary n_slots raz_game_summary.
 * Asynchronous on `stream`. */
typedef struct {
    uint64_t final_black, final_white;  /* the board when the game ended */
    uint32_t game_id, n_plies;
    uint8_t status;                     /* winner | flags, as raz_env_step */
    uint8_t resigned_black, resigned_white, enable_resign;
    uint32_t sims_lo;                   /* simulations run for this game so far (low 32 bits) */
} raz_game_summary;                     /* 32 bytes */
int raz_engine_records_extent(raz_engine* e, uint32_t first_slot, uint32_t n_slots, uint32_t* max_plies,
                              raz_stream_t stream);   /* synchronises `stream` */
int raz_engine_pack_records(raz_engine* e, uint32_t first_slot, uint32_t n_slots, uint32_t plies, void* d_headers,
                            uint32_t* d_root_n, raz_game_summary* d_summary, raz_stream_t stream);
/* Continuous batching.  The reference worker starts its next game the moment one ends (worker/self_play.py:95-137); a
 * lock-step batch idles every finished slot until its slowest game is over.  raz_engine_harvest, called between
 * raz_engine_step calls: every slot whose game has finished is emptied into the caller's OUTBOX (device arrays indexed by
 * game id - out_first_id, so the outbox is in id order whatever order the games finish in: headers out_games*max_plies*48
 * bytes, root_n out_games*max_plies*64 u32, summaries, done flags set to 1) and restarted - empty tree, fresh records - on
 * the next unplayed ids next_game_id, next_game_id + 1, ... (handed to the freed slots in slot order; at most n_new_ids of
 * them, the remaining freed slots idle).  sims_per_move[k] / resign_threshold[k] (nullable; NaN = no resignation rule;
 * NULL = the engine's run-time value) are the parameters of id next_game_id + k: a game keeps the threshold it was started
 * under, so its result depends on its id and parameters only - not on the batch size, the slot, or when other games end.
 * A finished game whose id has no outbox row stays in its slot (counted in `skipped`).  Not for series of games on a
 * carried tree (raz_engine_next_game).  Synchronises `stream`. */
typedef struct {
    uint32_t harvested;   /* games moved to the outbox by this call */
    uint32_t restarted;   /* of their slots, how many started a new game (ids next_game_id .. next_game_id + restarted - 1) */
    uint32_t skipped;     /* finished games left in place: id outside the outbox */
    uint32_t playing;     /* slots with a game in progress after the call */
} raz_harvest_result;
int raz_engine_harvest(raz_engine* e, uint32_t next_game_id, uint32_t n_new_ids, const uint32_t* sims_per_move,
                       const double* resign_threshold, uint32_t out_first_id, uint32_t out_games, void* d_headers,
                       uint32_t* d_root_n, raz_game_summary* d_summary, uint8_t* d_done, raz_harvest_result* result,
                       raz_stream_t stream);
/* Cross-game evaluation cache.  The reference evaluates every leaf through the net (agent/player.py:283-327 ->
 * agent/api.py:30-45).  With thousands of games on one device the same positions are asked for again and again (every game
 * passes through the same openings; a lock-step batch searches the same roots at the same time).  The net is a pure,
 * batch-invariant function of the position, so a repeated position is served from a table - bit-identical to evaluating
 * it again: games do not change, only the number of rows the net sees.  d_cache: caller-owned device buffer of
 * raz_leaf_cache_bytes(log2_entries, n_games * max(parallel_search_num, 1)) bytes, 256-byte aligned, cleared by this call;
 * NULL detaches.  max_discs: only positions with at most that many discs on the board are looked up and stored (0 = all;
 * positions deep in a game hardly ever repeat, the openings do).  Attach a FRESH (or re-attached = cleared) cache whenever the net's weights change.  With the split-f16
 * trunk (raz_net.reserved = 4) the rows still to evaluate are compacted, so the convolutions shrink with the hit rate;
 * other nets skip the served rows through their `active` mask.  stats: out4 = {hits, duplicates inside a batch, rows
 * evaluated, claims that found no room in the table} since the cache was attached (synchronises `stream`). */
size_t raz_leaf_cache_bytes(uint32_t log2_entries, size_t rows);
int raz_engine_set_leaf_cache(raz_engine* e, void* d_cache, size_t bytes, uint32_t log2_entries, uint32_t max_discs,
                              raz_stream_t stream);
int raz_engine_leaf_cache_stats(raz_engine* e, uint64_t* out4, raz_stream_t stream);
/* Statistics of the end-game solver's pool since raz_engine_start (csrc/raz_solver_pool.h; no reference counterpart - the reference
 * solves one position at a time, lib/alt/reversi_solver_cython.pyx:40-61).  out15:
 *   [0] [1] [2] shader-clock ticks the worker waves spent in their slow phases (memo traffic + task draws) / folding finished nodes into
 *               their parents / in their loops altogether
 *   [3] solves whose task tree was built            [4] answers published
 *   [5] rounds of the pool those solves were listed in before their answer
 *   [6] lane-iterations in which a worker lane searched a subtree      [7] wave-iterations executed (x 64 = lane slots)
 *   [8] subtrees finished      [9] subtrees drawn but not searched (their node was decided already)
 *   [10] worker-wave launches that found work
 *   [11] requests posted by all game slots, [12] most by one slot
 *   [13] rounds all slots' solves were listed in, [14] most of one slot (the batch's critical path in rounds of the pool).
 * Synchronises `stream`. */
int raz_engine_solver_stats(raz_engine* e, uint64_t* out15, raz_stream_t stream);
/* Diagnostics (no reference counterpart): `bytes` at `offset` of one of the engine's device arrays (raz_engine_device_ptr's numbering:
 * 3 the games' control blocks, 6 the solver blocks, 7 / 8 / 9 the solver pool's lane state / headers / active list) copied to host
 * memory.  RAZ_EINVAL when offset + bytes reaches beyond the array.  Synchronises the device. */
int raz_engine_debug_read(raz_engine* e, int which, size_t offset, size_t bytes, void* host_out);
/* config.play.resign_threshold is mutated while the worker runs (worker/self_play.py:250-260: +-0.01 per 100
 * no-resign test games); moves decided from the next raz_engine_step on use the new value.  Trees, records and
 * random streams are untouched. */
int raz_engine_set_resign_threshold(raz_engine* e, int has_threshold, double threshold);
/* The play_*.json text of ONE finished game, natively: the rows SelfPlayWorker.save_play_data appends for it
 * (worker/self_play.py:180-194: black.moves + white.moves; agent/player.py:166-179: 8 symmetric rows
 * [[own, enemy], [64 floats]] per searched ply, :357-364: z appended; saved policy per :132,366-385), as the
 * exact bytes json.dump would write for them - rows joined by ", ", WITHOUT the enclosing "[" "]" of the file
 * (a file is "[" + the fragments of its games joined by ", " + "]").  headers / root_n: n_plies records as
 * raz_engine_read_records returns them for that game; winner: status & 0x0f.  Host function, thread-safe.
 * Returns the number of bytes written (0 for a game without rows), negative on error (RAZ_ENOMEM: cap too
 * small - 8 x 2048 bytes per ply always suffice); *n_rows (nullable) receives the number of rows. */
long long raz_emit_game_rows_json(const void* headers, const uint32_t* root_n, int n_plies, int winner,
                                  int change_tau_turn, int save_policy_of_tau_1, char* out, size_t cap,
                                  int* n_rows);
/* float.__repr__(x) (the number format of json.dump) into out32 (>= 32 bytes, NUL-terminated); returns its length. */
int raz_format_float_repr(double x, char* out32);
/* Device pointers of the engine's arrays, for a caller that gathers / inspects them itself (e.g. RCCL): which = 0 ply headers, 1 root_n,
 * 2 root_w (NULL unless record_root_w), 3 the games' control blocks, 5 the phase profile, 6-9 the solver pool (diagnostics);
 * 10-14 the LEAF EXCHANGE with the net - the role of the reference's prediction queue (agent/player.py:329-346, agent/api.py:30-45):
 * one row per simulation slot (row = game * parallel_search_num + slot): 10 active u8 (1 = the last step's net forward evaluated the
 * row; rows served by the evaluation cache read 0), 11 own u64, 12 enemy u64 (the position as shown to the net, D4 transform
 * applied), 13 policy f32[64], 14 value f32 - what the net answered.  NULL for any other number. */
void* raz_engine_device_ptr(raz_engine* e, int which);

#ifdef __cplusplus
}
#endif
#endif /* RAZ_H */
