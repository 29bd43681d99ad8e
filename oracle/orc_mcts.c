/* orc_mcts.c — CPU oracle: one self-play game of the reference's MCTS player, restated in plain C.
 * TEST INFRASTRUCTURE (see orc.h).  Follows agent/player.py and worker/self_play.py statement by
 * statement for the reproducible mode (parallel_search_num = 1: one simulation in flight, so the
 * asyncio machinery :189-215,329-355 degenerates to a plain loop and virtual loss only leaves its
 * floating-point rounding behind).  Storage is literally the reference's: per key
 * (black, white, next_player) three 64-vectors N (f64), W (f64), P (f32 once the net wrote it).
 *
 * Random draws come from raz-rng-v1 (orc_rng.c) at the reference's four call sites; the net is
 * raznet-forward-v1 (orc_net.c).  Pinned by tests/golden/mcts_*.json, which are produced by the
 * UNMODIFIED reference player/worker code running with those two injected
 * (tests/golden/make_golden_mcts.py).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "orc.h"

/* ---------------- transposition storage (MCTSInfo, agent/player.py:62-66) ---------------- */
typedef struct {
    u64 black, white;
    int next_player;
    int used;
    double N[64], W[64];
    float P[64];
    uint8_t expanded[2]; /* per ReversiPlayer `expanded` set (:47, 325) */
} onode;

typedef struct {
    onode* nodes;
    size_t cap, count; /* open addressing, cap power of two */
} otable;

static u64 mix64(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
static size_t key_hash(u64 b, u64 w, int np) { return (size_t)mix64(b ^ mix64(w + 0x9e3779b97f4a7c15ULL * (u64)np)); }

static void table_init(otable* t) {
    t->cap = 1024;
    t->count = 0;
    t->nodes = (onode*)calloc(t->cap, sizeof(onode));
}
static void table_free(otable* t) { free(t->nodes); t->nodes = NULL; }

static onode* table_find(otable* t, u64 b, u64 w, int np) {
    size_t i = key_hash(b, w, np) & (t->cap - 1);
    while (t->nodes[i].used) {
        onode* n = &t->nodes[i];
        if (n->black == b && n->white == w && n->next_player == np) return n;
        i = (i + 1) & (t->cap - 1);
    }
    return NULL;
}
static onode* table_get(otable* t, u64 b, u64 w, int np); /* find-or-create (defaultdict access) */
static void table_grow(otable* t) {
    otable n;
    n.cap = t->cap * 2;
    n.count = 0;
    n.nodes = (onode*)calloc(n.cap, sizeof(onode));
    for (size_t i = 0; i < t->cap; ++i)
        if (t->nodes[i].used) {
            onode* d = table_get(&n, t->nodes[i].black, t->nodes[i].white, t->nodes[i].next_player);
            *d = t->nodes[i];
        }
    free(t->nodes);
    *t = n;
}
static onode* table_get(otable* t, u64 b, u64 w, int np) {
    onode* f = table_find(t, b, w, np);
    if (f) return f;
    if ((t->count + 1) * 2 > t->cap) table_grow(t);
    size_t i = key_hash(b, w, np) & (t->cap - 1);
    while (t->nodes[i].used) i = (i + 1) & (t->cap - 1);
    onode* n = &t->nodes[i];
    memset(n, 0, sizeof *n);
    n->black = b; n->white = w; n->next_player = np; n->used = 1;
    t->count++;
    return n;
}

/* ---------------- per-game state ---------------- */
typedef struct {
    const orc_play_cfg* cfg;
    const void* blob;
    size_t blob_bytes;
    uint32_t seed, game_id;
    uint32_t ev_expand, ev_choice, ev_dirichlet;
    otable tables[2];   /* tables[1] unused (aliased) when share_mtcs_info */
    int resigned[2];    /* ReversiPlayer.resigned (:58,125) */
    orc_solver* solver[2]; /* one ReversiSolver per player (:60, 435-436) */
    long long n_solved_leaves;
    long long n_sims, n_expand, n_mirror_hits, n_terminal;
} ogame;

static otable* player_table(ogame* g, int pl) { return (g->cfg->share_mtcs_info) ? &g->tables[0] : &g->tables[pl]; }

/* np.sum over a contiguous float32[64]: numpy's pairwise_sum for 8 <= n <= 128 keeps 8 running
 * partials r[j] += a[8i+j] and combines ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)). */
static float np_sum_f32_64(const float* a) {
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    for (int i = 8; i < 64; i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
}

/* var_q (:68-69) */
static double q_of(const onode* n, int a) { return n->W[a] / (n->N[a] + 1e-5); }

/* select_action_q_and_u (agent/player.py:395-428), dtype walk per SURVEY §8(a) P3 (numpy 2, NEP 50) */
static int select_action(ogame* g, int pl, const orc_env* env, int is_root) {
    const orc_play_cfg* c = g->cfg;
    otable* t = player_table(g, pl);
    onode* n = table_get(t, env->black, env->white, env->next_player);
    u64 legal = env->next_player == 1 ? orc_find_correct_moves(env->black, env->white)
                                      : orc_find_correct_moves(env->white, env->black);
    double sumN = 0.0;
    for (int i = 0; i < 64; ++i) sumN += n->N[i];
    double xx = sqrt(sumN);
    if (xx < 1.0) xx = 1.0; /* max(xx_, 1) */
    float p32[64];
    for (int i = 0; i < 64; ++i) p32[i] = n->P[i] * (float)((legal >> i) & 1);
    float sp = np_sum_f32_64(p32);
    if (sp > 0.0f) { /* temperature == 1 for every reachable turn: normalize(p, 1) in float32 */
        for (int i = 0; i < 64; ++i) p32[i] = p32[i] / sp;
    }
    double u[64];
    if (is_root && c->noise_eps > 0) {
        double noise[64];
        orc_dirichlet_noise_of_mask(legal, c->dirichlet_alpha, g->seed, g->game_id, g->ev_dirichlet++, noise);
        float keep = (float)(1.0 - c->noise_eps); /* python float is a weak scalar: stays float32 */
        for (int i = 0; i < 64; ++i) {
            double p64 = (double)(keep * p32[i]) + c->noise_eps * noise[i];
            u[i] = (c->c_puct * p64) * xx / (1.0 + n->N[i]);
        }
    } else {
        float cp = (float)c->c_puct; /* weak scalar * float32 array */
        for (int i = 0; i < 64; ++i) u[i] = ((double)(cp * p32[i])) * xx / (1.0 + n->N[i]);
    }
    int best = 0;
    double bestv = -1.0;
    for (int i = 0; i < 64; ++i) {
        double q = q_of(n, i);
        double v = (env->next_player == 1) ? (q + u[i] + 1000.0) : (-q + u[i] + 1000.0);
        v = v * (double)((legal >> i) & 1);
        if (i == 0 || v > bestv) { bestv = v; best = i; }
    }
    return best;
}

/* np.rot90(m, k=1) on an 8x8: out[i][j] = m[j][7-i];  np.flipud: out[i][j] = m[7-i][j] */
static void rot90_left(float* m) {
    float o[64];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) o[i * 8 + j] = m[j * 8 + (7 - i)];
    memcpy(m, o, sizeof o);
}
static void flipud(float* m) {
    float o[64];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) o[i * 8 + j] = m[(7 - i) * 8 + j];
    memcpy(m, o, sizeof o);
}

/* expand_and_evaluate (agent/player.py:283-327) */
static double expand_and_evaluate(ogame* g, int pl, const orc_env* env) {
    otable* t = player_table(g, pl);
    u64 black = env->black, white = env->white;
    double d[2];
    orc_rng_pair(g->seed, g->game_id, 0, g->ev_expand++, 0, 0, d);
    int is_flip = d[0] < 0.5;
    int rot = (int)(d[1] * 4);
    if (is_flip) { black = orc_flip_vertical(black); white = orc_flip_vertical(white); }
    for (int i = 0; i < rot; ++i) { black = orc_rotate90(black); white = orc_rotate90(white); }
    float pol[64], val;
    if (env->next_player == 1)
        orc_net_forward(g->blob, g->blob_bytes, black, white, pol, &val);
    else
        orc_net_forward(g->blob, g->blob_bytes, white, black, pol, &val);
    for (int i = 0; i < rot; ++i) rot90_left(pol);
    if (is_flip) flipud(pol);
    onode* n = table_get(t, env->black, env->white, env->next_player);
    memcpy(n->P, pol, sizeof pol);
    n->expanded[pl] = 1;
    onode* m = table_get(t, env->white, env->black, 3 - env->next_player); /* mirror key (:324) */
    memcpy(m->P, pol, sizeof pol);
    g->n_expand++;
    return (double)val; /* float(leaf_v) */
}

/* search_my_move (agent/player.py:217-281); returns leaf_v from the searching player's view */
static double search_my_move(ogame* g, int pl, orc_env* env, int is_root) {
    const orc_play_cfg* c = g->cfg;
    if (env->done) {
        g->n_terminal++;
        return env->winner == 1 ? 1.0 : (env->winner == 2 ? -1.0 : 0.0);
    }
    otable* t = player_table(g, pl);
    const u64 kb = env->black, kw = env->white;
    const int knp = env->next_player;
    if (c->use_solver_turn_in_simulation && env->turn >= c->use_solver_turn_in_simulation) { /* :237-251 */
        int action, score;
        int ok = orc_solver_solve(g->solver[pl], kb, kw, knp, 0, &action, &score);
        if (ok && action) { /* `if action:` — square 0 is ignored like None */
            if (knp != 1) score = -score;
            double leaf_v = score > 0 ? 1.0 : (score < 0 ? -1.0 : 0.0);
            onode* k = table_get(t, kb, kw, knp);
            k->N[action] += 1;
            k->W[action] += leaf_v;
            for (int i = 0; i < 64; ++i) k->P[i] = 0.0f;
            k->P[action] = 1.0f;
            onode* m2 = table_get(t, kw, kb, 3 - knp);
            m2->N[action] += 1;
            m2->W[action] -= leaf_v;
            for (int i = 0; i < 64; ++i) m2->P[i] = 0.0f;
            m2->P[action] = 1.0f;
            g->n_solved_leaves++;
            return leaf_v;
        }
    }
    onode* n = table_find(t, kb, kw, knp);
    if (!(n && n->expanded[pl])) {
        if (n && !c->share_mtcs_info) g->n_mirror_hits++; /* reached a key only mirror writes created */
        double leaf_v = expand_and_evaluate(g, pl, env);
        return knp == 1 ? leaf_v : -leaf_v;
    }
    double vl = (double)c->virtual_loss;
    double vlw = knp == 1 ? vl : -vl;
    int a = select_action(g, pl, env, is_root);
    orc_env_step(env, a);
    n = table_get(t, kb, kw, knp);
    n->N[a] += vl;
    n->W[a] -= vlw;
    double leaf_v = search_my_move(g, pl, env, 0);
    n = table_get(t, kb, kw, knp); /* table may have grown */
    n->N[a] += -vl + 1.0;
    n->W[a] += vlw + leaf_v;
    onode* m = table_get(t, kw, kb, 3 - knp);
    m->N[a] += 1.0;
    m->W[a] -= leaf_v;
    return leaf_v;
}

/* np.random.choice(range(64), p=policy) with the injected uniform (agent/player.py:112): cdf =
 * cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, u, side='right') */
static int choice64(const double* policy, double u) {
    double cdf[64], acc = 0.0;
    for (int i = 0; i < 64; ++i) { acc += policy[i]; cdf[i] = acc; }
    for (int i = 0; i < 64; ++i) cdf[i] = cdf[i] / acc;
    int idx = 0;
    while (idx < 64 && cdf[idx] <= u) ++idx;
    return idx < 64 ? idx : 63;
}

/* action_with_evaluation (agent/player.py:82-134).  Returns action (-1 = resign), fills rec. */
static int action_with_evaluation(ogame* g, int pl, u64 own, u64 enemy, int sims_per_move,
                                  int enable_resign, orc_ply_record* rec) {
    const orc_play_cfg* c = g->cfg;
    otable* t = player_table(g, pl);
    orc_env root;
    orc_env_update(&root, own, enemy, 1);
    const int turn = root.turn;
    memset(rec, 0, sizeof *rec);
    rec->own = root.black; /* after Board()'s `or default` quirk */
    rec->enemy = root.white;
    rec->turn = turn;
    if (c->use_solver_turn && turn >= c->use_solver_turn) { /* action_by_searching (:100-103, 150-161) */
        int a, score;
        if (orc_solver_solve(g->solver[pl], root.black, root.white, 1, 1, &a, &score)) {
            onode* n = table_get(t, root.black, root.white, 1);
            double sg = score > 0 ? 1.0 : (score < 0 ? -1.0 : 0.0);
            n->N[a] = 999;
            n->W[a] = sg * 999;
            for (int i = 0; i < 64; ++i) n->P[i] = 0.0f;
            n->P[a] = 1.0f;
            for (int i = 0; i < 64; ++i) { rec->root_n[i] = n->N[i]; rec->root_w[i] = n->W[i]; }
            rec->action = a;
            rec->solved = 1;
            rec->has_row = 0;
            rec->n = 999;
            rec->q = sg;
            return a;
        }
    }
    double policy[64];
    int action = 0;
    for (int tl = 0; tl < c->thinking_loop; ++tl) {
        if (turn > 0) {
            for (int s = 0; s < sims_per_move; ++s) { /* search_moves (:189-215) */
                orc_env env;
                orc_env_update(&env, own, enemy, 1);
                search_my_move(g, pl, &env, 1);
                g->n_sims++;
                rec->sims++;
            }
        } else { /* bypass_first_move (:143-148) */
            onode* n = table_get(t, root.black, root.white, 1);
            u64 legal = orc_find_correct_moves(root.black, root.white);
            int first = 0, cnt = orc_bit_count(legal);
            while (!((legal >> first) & 1)) ++first;
            n->N[first] = 1;
            n->W[first] = 0;
            for (int i = 0; i < 64; ++i) n->P[i] = (float)((double)((legal >> i) & 1) / (double)cnt);
        }
        onode* n = table_get(t, root.black, root.white, 1);
        /* calc_policy (:366-385) */
        if (turn < c->change_tau_turn) {
            double s = 0.0;
            for (int i = 0; i < 64; ++i) s += n->N[i];
            for (int i = 0; i < 64; ++i) policy[i] = n->N[i] / s;
        } else {
            int am = 0;
            for (int i = 1; i < 64; ++i) if (n->N[i] > n->N[am]) am = i;
            for (int i = 0; i < 64; ++i) policy[i] = 0.0;
            policy[am] = 1.0;
        }
        double d[2];
        orc_rng_pair(g->seed, g->game_id, 1, g->ev_choice++, 0, 0, d);
        action = choice64(policy, d[0]);
        int abv = 0; /* argmax(Q + (N>0)*100) (:113) */
        double bv = 0.0;
        for (int i = 0; i < 64; ++i) {
            double v = q_of(n, i) + (n->N[i] > 0 ? 100.0 : 0.0);
            if (i == 0 || v > bv) { bv = v; abv = i; }
        }
        double value_diff = q_of(n, action) - q_of(n, abv);
        rec->loops = tl + 1;
        if (turn <= c->start_rethinking_turn ||
            (value_diff > -0.01 && n->N[action] >= c->required_visit_to_decide_action))
            break;
    }
    onode* n = table_get(t, root.black, root.white, 1);
    for (int i = 0; i < 64; ++i) { rec->root_n[i] = n->N[i]; rec->root_w[i] = n->W[i]; }
    if (c->has_resign_threshold) { /* :123-130 */
        double mx = 0.0;
        for (int i = 0; i < 64; ++i) {
            double v = q_of(n, i) - (n->N[i] == 0 ? 10.0 : 0.0);
            if (i == 0 || v > mx) mx = v;
        }
        if (mx <= c->resign_threshold) {
            g->resigned[pl] = 1;
            if (enable_resign && turn >= c->allowed_resign_turn) {
                rec->action = -1;
                rec->has_row = 0;
                return -1;
            }
        }
    }
    if (c->save_policy_of_tau_1) { /* :132, calc_policy_by_tau_1 */
        double s = 0.0;
        for (int i = 0; i < 64; ++i) s += n->N[i];
        for (int i = 0; i < 64; ++i) rec->saved_policy[i] = n->N[i] / s;
    } else
        memcpy(rec->saved_policy, policy, sizeof policy);
    rec->has_row = 1;
    rec->action = action;
    rec->n = n->N[action];
    rec->q = q_of(n, action);
    return action;
}

/* SelfPlayWorker.start_game (worker/self_play.py:139-175) for one game; plies[] needs room for
 * max_plies records.  Returns the number of plies recorded, or -1 on error. */
int orc_selfplay_game(const orc_play_cfg* cfg, const void* blob, size_t blob_bytes, uint32_t seed,
                      uint32_t game_id, int sims_per_move, orc_ply_record* plies, int max_plies,
                      orc_game_summary* sum) {
    ogame g;
    memset(&g, 0, sizeof g);
    g.cfg = cfg; g.blob = blob; g.blob_bytes = blob_bytes; g.seed = seed; g.game_id = game_id;
    if (cfg->parallel_search_num != 1) return -1;
    g.solver[0] = orc_solver_new();
    g.solver[1] = orc_solver_new();
    table_init(&g.tables[0]);
    if (!cfg->share_mtcs_info) table_init(&g.tables[1]);
    double d[2];
    orc_rng_pair(seed, game_id, 3, 0, 0, 0, d);
    int enable_resign = cfg->disable_resignation_rate <= d[0]; /* self_play.py:144 */
    orc_env env;
    orc_env_reset(&env);
    int np = 0;
    while (!env.done) {
        if (np >= max_plies) { np = -1; break; }
        int pl = env.next_player == 1 ? 0 : 1;
        u64 own = pl == 0 ? env.black : env.white, enemy = pl == 0 ? env.white : env.black;
        int a = action_with_evaluation(&g, pl, own, enemy, sims_per_move, enable_resign, &plies[np]);
        plies[np].player = env.next_player;
        ++np;
        orc_env_step(&env, a);
    }
    if (sum) {
        memset(sum, 0, sizeof *sum);
        sum->winner = env.winner; sum->black = env.black; sum->white = env.white; sum->turn = env.turn;
        sum->plies = np; sum->enable_resign = enable_resign;
        sum->resigned_black = g.resigned[0]; sum->resigned_white = g.resigned[1];
        sum->drop_draw_u = d[1];
        sum->n_sims = g.n_sims; sum->n_expand = g.n_expand; sum->n_mirror_hits = g.n_mirror_hits;
        sum->n_terminal = g.n_terminal;
        sum->n_nodes = (long long)(g.tables[0].count + (cfg->share_mtcs_info ? 0 : g.tables[1].count));
        sum->n_solved_leaves = g.n_solved_leaves;
        sum->n_solver_nodes = orc_solver_nodes(g.solver[0]) + orc_solver_nodes(g.solver[1]);
    }
    orc_solver_free(g.solver[0]);
    orc_solver_free(g.solver[1]);
    table_free(&g.tables[0]);
    if (!cfg->share_mtcs_info) table_free(&g.tables[1]);
    return np;
}
