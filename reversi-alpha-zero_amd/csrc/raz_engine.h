// raz_engine.h — device-resident state of the batched self-play engine (one wavefront per game).
//
// Reference objects this replaces, per game:  SelfPlayWorker.start_game's loop state
// (worker/self_play.py:139-175), two ReversiPlayer instances (agent/player.py:28-61) and their
// MCTSInfo dictionaries var_n / var_w / var_p keyed by (black, white, next_player) (:18,62-66).
//
// HBM layout (all arrays indexed by game slot g first, so one game's data is contiguous for the
// wave that owns it and different games never share a cache line of mutable data):
//   tree nodes   : COMPACT, variable-size: a node holds statistics for its position's L legal moves only (average 8.5 of
//                  64 squares).  Node = 40-byte header {black, white, legal, tag, mirror link, creation index} followed by
//                  four arrays of L entries - W f64 | N u32 | P f32 | child u32 - entry r belonging to the r-th legal
//                  square in ascending order: 40 + 20 L bytes rounded up to 8 (~210 B on average; the reference's
//                  per-key f64[64] x 3 is 1536 B, the fixed 64-lane block this replaces was 1408 B).  A game's nodes
//                  are packed back to back in its byte pool.  A node is referred to by a LINK = (offset / 8) << 6 | L,
//                  so whoever follows a link knows the node's size and array offsets before its header arrives: a
//                  descent still costs ONE dependent memory round trip per level (header + the four arrays requested
//                  together, lane r < L loading entry r).  P holds the prior already masked by the legal moves and
//                  normalised (what select_action_q_and_u recomputes at every visit, player.py:404-413); child[r] is
//                  the link of the node reached by the r-th legal move (0 = not linked yet, bit 31 = edge known to end
//                  the game, low bits = winner).  PUCT runs in RANK space (lane r = r-th legal move): ascending rank
//                  is ascending square, so numpy's first-maximum argmax and the Dirichlet sample order are unchanged.
//   node dir     : link of the i-th node created in a game (creation order = ascending offset): k_gc walks it
//   hash table   : H slots of 32 B {black, white, idx|tag}, open addressing, linear probing, probed
//                  16 slots (512 B, one coalesced request) at a time; used only for first arrivals
//   per-sim path : 64 x (node idx u32, action|np u8)
//   records      : per ply 48 B header + root N u32 x64 (+ optional root W f64 x64)
#pragma once
#include <stdint.h>
#include "../../include/raz.h"

#define RAZ_NODE_HDR_BYTES 40
#define RAZ_NODE_ENTRY_BYTES 20          // W f64 + N u32 + P f32 + child u32 per legal move
#define RAZ_NODE_MAX_BYTES 704           // 40 + 20 x 33 legal moves (the most a Reversi position can have), rounded up to 8
#define RAZ_NODE_DEFAULT_BYTES 232       // default pool budget per node of nodes_per_game (the whole-game average is ~212)
#define RAZ_NODE_OUT_BYTES 1408          // staging of raz_engine_read_node: the node expanded to W f64 x64 | N u32 x64 | P f32 x64
#define RAZ_SLOT_BYTES 32
#define RAZ_PROBE 16          // slots the solver memo reads per request
// The tree's hash table is probed in ALIGNED groups of 4 slots = one 128-byte line per request (round 4; 16 unaligned slots = 512 B
// before): group (hash & ~3), then the next group.  A key sits at the first free slot of that order, so a lookup meets it before
// it meets an empty slot; at the table's load of <= 0.5 the first group decides 9 requests of 10.
#define RAZ_TABLE_PROBE 4

#define RAZ_LEAF_NONE 0
#define RAZ_LEAF_EXPAND 1
#define RAZ_LEAF_TERMINAL 2
#define RAZ_LEAF_SOLVED 3   // the in-simulation solver answered (agent/player.py:237-251)
#define RAZ_LEAF_PARKED 4   // parallel_search_num > 1: the key is in now_expanding (agent/player.py:253-254)
#define RAZ_LEAF_SOLVE_PENDING 5   // the descent stands at an in-simulation solve that ran out of the launch's solver budget: leaf_node = the
                                   // node to go on from, depth = its level, leaf_action = 1 + the rank of the edge already chosen there (0: none)

// State of one of the parallel_search_num simulation slots of a game (k_tree_par)
#define RAZ_SIM_FREE 0
#define RAZ_SIM_WAIT_NET 1     // its leaf is in the prediction queue (expand_and_evaluate awaits the future)
#define RAZ_SIM_WAIT_EXPAND 2  // sleeping on now_expanding at node sim_parked
#define RAZ_SIM_SOLVING 3      // its descent is suspended at an in-simulation solve (RAZ_LEAF_SOLVE_PENDING in its block)

// End-game solver, POOLED across games (raz_solver_pool.h).  A tree kernel that needs f_mode(position) for a position with 5..14
// empties (more than RAZ_SOLVER_SCALAR_EMPTIES = 4) POSTS a request in its game's block of E.solver_ws (header below, then the task tree: three plies + a fourth of tasks) and suspends where it
// stands; between tree launches k_solve_scan builds the task trees of new requests and folds finished tasks into answers, and
// k_solve_run - a fixed pool of worker waves that belongs to no game - hands ONE subtree to every LANE, whatever game it comes from.
#define RAZ_SOLVE_IDLE 0u
#define RAZ_SOLVE_REQUESTED 1u   // posted by a tree kernel: own0 / enemy0 / exact are valid, gen was bumped
#define RAZ_SOLVE_RUNNING 2u     // its task tree is built; workers take tasks off `next`
#define RAZ_SOLVE_ANSWERED 3u    // the state word carries f(own0, enemy0): stays until another position is requested
// The state word: bits 0-7 the state; an ANSWERED word also holds the answer - bits 8-15 RAZ_SOLVE_DONE / RAZ_SOLVE_NONE, 16-23 move + 1,
// 24-31 score + 128 - so that ONE 4-byte store publishes it: the pool's round may run on its own stream beside the tree kernels of the
// following steps (raz_engine.hip, pool_every), and a tree kernel that sees ANSWERED must see the whole answer.
#define RAZ_SOLVE_STATE(w) ((w) & 0xffu)
#define RAZ_SOLVE_ANSWER_WORD(kind, move, score) \
    (RAZ_SOLVE_ANSWERED | ((uint32_t)(kind) << 8) | ((uint32_t)(((move) + 1) & 0xff) << 16) | ((uint32_t)(((score) + 128) & 0xff) << 24))
#define RAZ_SOLVE_ANSWER_KIND(w) (((w) >> 8) & 0xffu)
#define RAZ_SOLVE_ANSWER_MOVE(w) ((int)(((w) >> 16) & 0xffu) - 1)
#define RAZ_SOLVE_ANSWER_SCORE(w) ((int)(((w) >> 24) & 0xffu) - 128)
struct raz_solve_hdr {           // 64 bytes at the start of a game's solver block
    uint32_t state, gen;
    unsigned long long own0, enemy0;
    uint32_t exact;
    uint32_t k_n2, tasks, total; // root moves | level-2 nodes << 8; level-3 nodes; subtrees the workers search (the tasks)
    uint32_t next;               // next task to hand out (workers: atomicAdd)
    int32_t ans_move;            // (a copy of what the state word carries: diagnostics)
    uint32_t limit;              // tasks [0, limit) may be handed out in this round (k_solve_scan: every task of an exact solve; of a win/loss
                                 // solve the tasks below the first RAZ_SOLVER_NE_WINDOW root moves that are still open)
    uint32_t posted;             // requests this game slot has posted since raz_engine_start (the tree kernels' word; statistics)
    uint32_t rounds;             // rounds of the pool the solve has been listed in (statistics)
    uint32_t rounds_total;       // ... and all solves of this game slot since raz_engine_start
};
#ifdef __cplusplus
static_assert(sizeof(raz_solve_hdr) == 64, "raz_solve_hdr layout");
static_assert(__builtin_offsetof(raz_solve_hdr, state) == 0 && __builtin_offsetof(raz_solve_hdr, gen) == 4, "{state, gen} is ONE aligned 8-byte word (raz_engine_core.h solver_solve)");
#endif
#define RAZ_SOLVER_TREE_BYTES 12288   // >= sizeof(SolverTree) (raz_solver_pool.h, checked there): the top three plies, folded in LDS
#define RAZ_SOLVER_DEEP_BYTES 135168  // >= sizeof(SolverDeep): the positions three plies down and, below the larger ones, a fourth ply of tasks
#define RAZ_SOLVER_WS_BYTES (64 + RAZ_SOLVER_TREE_BYTES + RAZ_SOLVER_DEEP_BYTES)
// a worker wave of the pool: 16 words of lane state (the search in hand and the task drawn ahead) and 14 frames of 32 B per lane
#define RAZ_SOLVER_WORKER_STATE_BYTES (16 * 64 * 8)
#define RAZ_SOLVER_WORKER_FRAME_BYTES (14 * 64 * 32)
struct raz_solver_pool_hdr {     // one per slice of the batch (64 bytes)
    uint32_t n_active;           // solves with tasks left to hand out, listed in `active`
    uint32_t unused;
    uint32_t pad[14];
};
#define RAZ_PHASE_NEW_MOVE 0
#define RAZ_PHASE_SEARCH 1
#define RAZ_PHASE_DONE 2
#define RAZ_PHASE_IDLE 3

#define RAZ_ERR_POOL_FULL 1u
#define RAZ_ERR_TABLE_FULL 2u
#define RAZ_ERR_RECORDS_FULL 4u
#define RAZ_ERR_PATH_FULL 8u

// Hash-table slot: key -> node index.  Only consulted when a position is reached for the first time
// along an edge (afterwards the parent's child link leads straight to the node).
struct raz_slot {  // 32 bytes
    unsigned long long black, white;
    uint32_t idx_tag;  // used<<7 | owner<<2 | next_player   (the solver memo packs its answer into the upper bits)
    uint32_t link;     // the node's link
    unsigned long long pad1;
};
#define RAZ_SLOT_USED 0x80u
#define RAZ_SLOT_KEYMASK 0x07u

// Header at the start of a node (one 40-byte broadcast load, requested together with the node's arrays).
struct raz_node_hdr {
    unsigned long long black, white;  // key
    unsigned long long legal;         // legal moves of the side to move (computed once); L = popcount
    uint32_t tag;                     // next_player | owner<<2 | expanded_by_black<<4 | expanded_by_white<<5
                                      // | being expanded by black<<6 / white<<7 (now_expanding, parallel_search_num > 1)
    uint32_t mirror;                  // link of the colour-mirrored key's node, 0xffffffff = none yet
    uint32_t index;                   // creation index in the game's node directory
    uint32_t gc_index;                // scratch of k_gc: the index after compaction
};
// A LINK names a node: bits 6..30 = byte offset in the game's pool / 8, bits 0..5 = L (legal moves = array length).
// 0 = no link (offset 0 of a pool is never a node), bit 31 = not a node (RAZ_NO_NODE, or a terminal edge in child[]).
#define RAZ_LINK_MAX_UNITS (1u << 25)    // pools of up to 256 MB per game
#define RAZ_NO_NODE 0xffffffffu

struct raz_ply_header {  // 48 bytes, one per recorded ply (== orc_ply_record minus the vectors)
    unsigned long long own, enemy;  // mover's view, as ReversiPlayer.action_with_evaluation gets them
    double n, q;                    // ActionWithEvaluation.n / .q
    int8_t action;                  // 0..63, -1 = resigned
    uint8_t player;                 // 1 black / 2 white to move
    uint8_t turn;
    uint8_t has_row;                // a training row is emitted for this ply
    uint32_t sims;                  // simulations run for this move (all thinking loops)
    uint32_t loops;
    uint32_t flags;                 // bit 0: move chosen by the exact solver (action_by_searching)
};

// One game slot's control state: 64 dwords.  k_tree loads it with one coalesced request (lane i =
// dword i), keeps it in a single VGPR for the whole launch and stores it back once; the other
// kernels and the host address the fields by name.
struct raz_game {
    unsigned long long root_black, root_white;      // the real board (SelfPlayWorker's env)
    unsigned long long leaf_b, leaf_w, leaf_legal;  // in-flight leaf: position key (searching player's view), legal moves
    unsigned long long sims, leaves, selections;    // statistics: simulations, leaves sent to the net, PUCT selections
    uint32_t game_id, player, status, phase;        // player 1 black / 2 white to move; status = raz_env_step's
    uint32_t enable_resign, resigned[2], one_move;  // resigned[p]: ReversiPlayer.resigned of black / white
    uint32_t ev_expand, ev_choice, ev_dirichlet, sims_per_move;  // raz-rng-v1 event counters
    int32_t sims_left;
    uint32_t loops_done, move_sims, pool_used;      // pool_used: next free offset of the node pool in 8-byte units (starts at 1)
    uint32_t n_plies, error, root_node, leaf_kind;
    uint32_t leaf_sym, leaf_np, depth, leaf_action; // D4 transform shown to the net, side to move at the leaf, path length
    uint32_t leaf_node, leaf_slot, leaf_tag, leaf_mirror;  // existing node of the leaf (or RAZ_NO_NODE) / empty slot found
    uint32_t leaf_term_v;                           // f32 bits: value of a terminal / solved leaf
    // parallel_search_num > 1 (k_tree_par).  Every simulation slot owns a 64-dword block with THIS layout in
    // which only the in-flight-simulation fields (leaf_*, depth, sim_*) are meaningful, so that moving a slot
    // into / out of the register-resident control block is one masked select / one masked store.
    uint32_t sim_state, sim_seq, sim_parked;        // per slot: RAZ_SIM_*, order number (queue put order / sleep order), node slept on
    uint32_t par_seq_next, par_stage;               // per game: next order number; 0 = round boundary, 1 = filling (C), 2 = refilling (C'), 4 = polling sleepers (D: only while a solve is suspended)
    // continuous batching (raz_engine_harvest): a game started by a harvest carries the resign threshold it was started
    // under, so that its result does not depend on when other games finish (worker/self_play.py:250-260 mutates the value)
    unsigned long long resign_thr;                  // f64 bits, valid when resign_mode == 1
    uint32_t resign_mode;                           // 0 = the engine's run-time value (raz_engine_set_resign_threshold), 1 = resign_thr, 2 = no rule
    uint32_t node_count;                            // nodes in the pool (= entries of the node directory)
    uint32_t par_dmask;                             // stage D suspended by a solve: the sleepers still to poll
    uint32_t pad[9];
};
#ifdef __cplusplus
static_assert(sizeof(raz_game) == 256, "raz_game must be 64 dwords");
#endif

// Cross-game evaluation cache (raz_leaf_cache.hip): pointers into the caller-provided cache buffer.  Passed BY VALUE to kernels.
struct raz_leaf_cache_dev {
    unsigned long long* tags;      // [E] 0 = empty, else hash64(own, enemy) | 1
    unsigned long long* keys;      // [E][2] the full key
    uint32_t *stamp, *owner, *ready;   // [E] step that claimed the entry / exchange row computing it / answer present
    float* pv;                     // [E][72] policy 64, value, pad
    unsigned long long* counters;  // [0] hits, [1] in-batch duplicates, [2] rows evaluated, [3] claims that found no room
    uint32_t* n_compact;           // [slice] rows in the slice's compact list
    uint32_t *list, *role;         // [rows] compact list (per slice, at the slice's first row) / what each row is doing this step
    uint32_t mask, entries;
    uint32_t max_discs;            // only positions with at most this many discs are cached (later positions hardly ever repeat)
};

// Pointers into the caller-provided workspace + the play parameters.  Passed BY VALUE to kernels.
struct raz_engine_dev {
    raz_engine_config cfg;
    uint32_t B, C, H, max_plies;   // C: node COUNT capacity per game (directory / table sizing)
    unsigned long long pool_bytes; // bytes of one game's node pool (multiple of 8, <= 8 x RAZ_LINK_MAX_UNITS)
    uint32_t xk;                   // 1: the solver pool's round runs on its own stream BESIDE tree launches (pool_every > 1): the words the two sides
                                   // exchange are agent-scope (raz_engine_core.h xk_*).  0: every hand-off crosses a kernel boundary - plain accesses
    uint32_t K;                    // simulation slots per game: parallel_search_num (1 for the classic one-in-flight kernel)
    uint32_t par;                  // 1: k_tree_par drives the games (per-slot state in `sim`), 0: k_tree
    uint32_t* sim;                 // [B][K][64] slot blocks (raz_game layout), par only
    raz_game* game;                // [B]
    // leaf exchange with the net kernel, and the path of the simulation in flight; one entry per
    // simulation slot: index g * K + slot
    uint8_t* nn_active;            // [B*K]
    unsigned long long *nn_own, *nn_enemy;
    float *nn_policy /*[B*K][64]*/, *nn_value;
    uint32_t *path_node /*[B*K][64]*/, *path_mirror /*[B*K][64]*/;
    uint16_t* path_act /*[B*K][64]: action | side to move << 6 | rank of the action among the node's legal moves << 8*/;
    // tree
    raz_slot* table;               // [B][H]
    unsigned char* nodes;          // [B][pool_bytes]
    uint32_t* node_dir;            // [B][C] link of the i-th node created
    // records
    raz_ply_header* rec;           // [B][max_plies]
    uint32_t* rec_n;               // [B][max_plies][64]
    double* rec_w;                 // [B][max_plies][64] or NULL
    // reduced by k_stats: [0] finished games, [1] total sims, [2] error flags, [3] nn leaves, [4] selections, [5] max pool_used over live games, [6] idle or finished slots
    raz_slot* memo;                // [B][M] solved positions: {own, enemy, used<<31 | exact<<30 | (move+1)<<8 | score+128}
    uint32_t M;
    unsigned char* solver_ws;      // [B][RAZ_SOLVER_WS_BYTES] per game: raz_solve_hdr + the task tree of the solve in flight (SolverTree, SolverDeep)
    // the worker pool of the end-game solver (raz_solver_pool.h): W waves, split evenly over the slices of the batch
    uint32_t W;                    // worker waves
    raz_solver_pool_hdr* pool_hdr; // [kMaxParts]
    uint32_t* pool_active;         // [B] slice h lists its active solves (game slots) from its first game's index on
    unsigned long long* pool_state;   // [W][8][64] parked lane state
    unsigned long long* pool_frames;  // [W][14][64][4] the lanes' DFS frames
    unsigned char* node_out;       // RAZ_NODE_OUT_BYTES + 64: staging of raz_engine_read_node
    uint32_t* gc_remap;            // [B][C] creation index -> link after compaction, during k_gc
    unsigned long long* counters;  // ... [20..27] the solver pool's statistics (raz_engine_solver_stats)
    unsigned long long* prof;      // [B][8] optional phase profile (cfg.reserved & 1)
};
