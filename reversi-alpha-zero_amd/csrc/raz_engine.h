// raz_engine.h — device-resident state of the batched self-play engine (one wavefront per game).
//
// Reference objects this replaces, per game:  SelfPlayWorker.start_game's loop state
// (worker/self_play.py:139-175), two ReversiPlayer instances (agent/player.py:28-61) and their
// MCTSInfo dictionaries var_n / var_w / var_p keyed by (black, white, next_player) (:18,62-66).
//
// HBM layout (all arrays indexed by game slot g first, so one game's data is contiguous for the
// wave that owns it and different games never share a cache line of mutable data):
//   tree nodes   : node g,i = 1 KiB block  [W f64 x64 | N u32 x64 | P f32 x64]; P holds the prior
//                  already masked by the legal moves and normalised (what select_action_q_and_u
//                  recomputes at every visit, player.py:404-413, depends only on the node)
//   hash table   : H slots of 32 B {black, white, legal, idx|tag, mirror}, open addressing, linear
//                  probing, probed 16 slots (512 B, one coalesced request) at a time
//   per-sim path : 64 x (node idx u32, slot idx u32, action|np u8)
//   records      : per ply 48 B header + root N u32 x64 (+ optional root W f64 x64)
#pragma once
#include <stdint.h>
#include "../../include/raz.h"

#define RAZ_NODE_BYTES 1024
#define RAZ_SLOT_BYTES 32
#define RAZ_PROBE 16

#define RAZ_LEAF_NONE 0
#define RAZ_LEAF_EXPAND 1
#define RAZ_LEAF_TERMINAL 2

#define RAZ_PHASE_NEW_MOVE 0
#define RAZ_PHASE_SEARCH 1
#define RAZ_PHASE_DONE 2
#define RAZ_PHASE_IDLE 3

#define RAZ_ERR_POOL_FULL 1u
#define RAZ_ERR_TABLE_FULL 2u
#define RAZ_ERR_RECORDS_FULL 4u
#define RAZ_ERR_PATH_FULL 8u

// A table slot doubles as the node header: key, the mover's legal-move mask (computed once, when the
// position is first reached), the node index and flags, and the node index of the colour-mirrored key.
struct raz_slot {  // 32 bytes
    unsigned long long black, white;
    unsigned long long legal;
    uint32_t idx_tag;  // node index << 8 | used<<7 | expanded_by_white<<5 | expanded_by_black<<4 | owner<<2 | next_player
    uint32_t mirror;   // node index of the mirrored key (player.py:391-393), 0xffffffff = not looked up yet
};
#define RAZ_SLOT_USED 0x80u
#define RAZ_SLOT_KEYMASK 0x07u

struct raz_ply_header {  // 48 bytes, one per recorded ply (== orc_ply_record minus the vectors)
    unsigned long long own, enemy;  // mover's view, as ReversiPlayer.action_with_evaluation gets them
    double n, q;                    // ActionWithEvaluation.n / .q
    int8_t action;                  // 0..63, -1 = resigned
    uint8_t player;                 // 1 black / 2 white to move
    uint8_t turn;
    uint8_t has_row;                // a training row is emitted for this ply
    uint32_t sims;                  // simulations run for this move (all thinking loops)
    uint32_t loops;
    uint32_t pad;
};

// Pointers into the caller-provided workspace + the play parameters.  Passed BY VALUE to kernels.
struct raz_engine_dev {
    raz_engine_config cfg;
    uint32_t B, C, H, max_plies;
    // game state
    unsigned long long *root_black, *root_white;
    uint8_t *g_player, *g_status, *g_phase, *g_enable_resign, *g_resigned /*[B][2]*/;
    uint32_t *g_game_id, *ev_expand, *ev_choice, *ev_dirichlet;
    uint32_t *sims_per_move, *loops_done, *move_sims, *pool_used, *n_plies, *g_error;
    int32_t* sims_left;
    unsigned long long* g_sims;  // simulations executed, per game
    unsigned long long *g_leaves, *g_selections;  // leaves sent to the net / PUCT selections, per game
    // in-flight simulation
    uint8_t *leaf_kind, *leaf_sym, *leaf_np, *depth, *nn_active;
    unsigned long long *leaf_b, *leaf_w, *leaf_legal, *nn_own, *nn_enemy;
    float *leaf_term_v, *nn_policy /*[B][64]*/, *nn_value;
    uint32_t *path_node /*[B][64]*/, *path_slot /*[B][64]*/;
    uint8_t* path_act /*[B][64]*/;
    // tree
    raz_slot* table;               // [B][H]
    unsigned char* nodes;          // [B][C][1024]
    // records
    raz_ply_header* rec;           // [B][max_plies]
    uint32_t* rec_n;               // [B][max_plies][64]
    double* rec_w;                 // [B][max_plies][64] or NULL
    // reduced by k_stats: [0] finished games, [1] total sims, [2] error flags, [3] nn leaves, [4] selections
    unsigned long long* counters;
};
