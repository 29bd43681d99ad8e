/* orc_mcts.c — CPU oracle: one self-play game of the reference's MCTS player, restated in plain C.
 * TEST INFRASTRUCTURE (see orc.h).  Follows agent/player.py and worker/self_play.py statement by
 * statement.  parallel_search_num = 1 is the reference's reproducible mode (one simulation in
 * flight: the asyncio machinery :189-215,329-355 degenerates to a plain loop and virtual loss only
 * leaves its floating-point rounding behind); parallel_search_num > 1 follows the timer-free round
 * schedule raz-sched-v1 defined at search_moves() below.  Storage is literally the reference's: per key
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
    uint8_t expanding[2]; /* per ReversiPlayer `now_expanding` set (:48, 294, 326) */
    uint8_t has_p;        /* key in var_p (what a new ReversiPlayer on a used MCTSInfo takes as `expanded`, :47) */
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
    orc_nn_fn nn;       /* nullable: the NN seam (ReversiPlayer(api=...), agent/player.py:41,346) injected by the test */
    void* nn_ctx;
    uint32_t seed, game_id;
    uint32_t ev_expand, ev_choice, ev_dirichlet;
    otable tables[2];   /* tables[1] unused (aliased) when share_mtcs_info */
    int resigned[2];    /* ReversiPlayer.resigned (:58,125) */
    orc_solver* solver[2]; /* one ReversiSolver per player (:60, 435-436) */
    long long n_solved_leaves;
    long long n_sims, n_expand, n_mirror_hits, n_terminal, n_parked;
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

/* ---------------- simulations in flight (agent/player.py:189-355) ----------------
 * The reference runs `simulation_num_per_move` coroutines of search_my_move under
 * asyncio.Semaphore(parallel_search_num) beside a prediction_worker coroutine that drains the
 * prediction queue into ONE api.predict call (:329-349).  Here a simulation is an explicit record
 * (the coroutine's frame: path so far, position, pending net answer) and the event loop is the
 * round schedule raz-sched-v1 (search_moves below).  With parallel_search_num = 1 this is exactly
 * the recursion of :217-281 unrolled - pinned bit for bit by tests/golden/mcts_games.json. */
#define ORC_MAX_PAR 16 /* prediction_queue_size (config.py:141): put() would block beyond it */
enum { SIM_FREE = 0, SIM_WAIT_NET = 1, SIM_WAIT_EXPAND = 2 };
typedef struct {
    int state;
    int depth;
    u64 kb[64], kw[64];   /* keys along the path (the frames of the recursion) */
    int knp[64], act[64];
    orc_env env;          /* position at the frontier */
    float pol[64], val;   /* the net's answer for the leaf being expanded, policy already un-transformed */
} osim;

/* "on returning search path" (:273-281) for every frame of the simulation, deepest first */
static void backup_path(ogame* g, int pl, const osim* s, double leaf_v) {
    otable* t = player_table(g, pl);
    const double vl = (double)g->cfg->virtual_loss;
    for (int d = s->depth - 1; d >= 0; --d) {
        const int a = s->act[d];
        const double vlw = s->knp[d] == 1 ? vl : -vl;
        onode* n = table_get(t, s->kb[d], s->kw[d], s->knp[d]);
        n->N[a] += -vl + 1.0;
        n->W[a] += vlw + leaf_v;
        onode* m = table_get(t, s->kw[d], s->kb[d], 3 - s->knp[d]);
        m->N[a] += 1.0;
        m->W[a] -= leaf_v;
    }
}

/* expand_and_evaluate (:283-327) up to `await future`: now_expanding.add(key), the random D4, the
 * request to the net (its answer is position-determined, so it is computed here and kept in the
 * simulation's record until the event loop resumes it) */
static void expand_begin(ogame* g, int pl, osim* s) {
    otable* t = player_table(g, pl);
    const orc_env* env = &s->env;
    u64 black = env->black, white = env->white;
    onode* n = table_get(t, env->black, env->white, env->next_player);
    n->expanding[pl] = 1;
    double d[2];
    orc_rng_pair(g->seed, g->game_id, 0, g->ev_expand++, 0, 0, d);
    int is_flip = d[0] < 0.5;
    int rot = (int)(d[1] * 4);
    if (is_flip) { black = orc_flip_vertical(black); white = orc_flip_vertical(white); }
    for (int i = 0; i < rot; ++i) { black = orc_rotate90(black); white = orc_rotate90(white); }
    const u64 o_ = env->next_player == 1 ? black : white, e_ = env->next_player == 1 ? white : black;
    if (g->nn) g->nn(g->nn_ctx, o_, e_, s->pol, &s->val);   /* injected api.predict (same seam as the reference's) */
    else orc_net_forward(g->blob, g->blob_bytes, o_, e_, s->pol, &s->val);
    for (int i = 0; i < rot; ++i) rot90_left(s->pol);
    if (is_flip) flipud(s->pol);
}

/* ... and after it: var_p[key] = var_p[another_side_key] = leaf_p, expanded.add, now_expanding.remove,
 * then the returns of search_my_move up the path */
static void expand_finish(ogame* g, int pl, osim* s) {
    otable* t = player_table(g, pl);
    const orc_env* env = &s->env;
    onode* n = table_get(t, env->black, env->white, env->next_player);
    memcpy(n->P, s->pol, sizeof s->pol);
    n->has_p = 1;
    n->expanded[pl] = 1;
    n->expanding[pl] = 0;
    onode* m = table_get(t, env->white, env->black, 3 - env->next_player); /* mirror key (:324) */
    memcpy(m->P, s->pol, sizeof s->pol);
    m->has_p = 1;
    g->n_expand++;
    double leaf_v = (double)s->val; /* float(leaf_v) */
    backup_path(g, pl, s, env->next_player == 1 ? leaf_v : -leaf_v);
}

/* search_my_move (:217-281) from the simulation's frontier until it finishes (returns SIM_FREE,
 * statistics backed up) or blocks: SIM_WAIT_NET (a leaf went to the net) / SIM_WAIT_EXPAND (the
 * key is in now_expanding, :253-254) */
static int descend(ogame* g, int pl, osim* s, int polling) {
    const orc_play_cfg* c = g->cfg;
    otable* t = player_table(g, pl);
    orc_env* env = &s->env;
    for (;;) {
        if (env->done) {
            g->n_terminal++;
            backup_path(g, pl, s, env->winner == 1 ? 1.0 : (env->winner == 2 ? -1.0 : 0.0));
            return SIM_FREE;
        }
        const u64 kb = env->black, kw = env->white;
        const int knp = env->next_player;
        /* a simulation woken from the now_expanding sleep stands behind the solver look already */
        if (!polling && c->use_solver_turn_in_simulation && env->turn >= c->use_solver_turn_in_simulation) { /* :237-251 */
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
                k->has_p = 1;
                onode* m2 = table_get(t, kw, kb, 3 - knp);
                m2->N[action] += 1;
                m2->W[action] -= leaf_v;
                for (int i = 0; i < 64; ++i) m2->P[i] = 0.0f;
                m2->P[action] = 1.0f;
                m2->has_p = 1;
                g->n_solved_leaves++;
                backup_path(g, pl, s, leaf_v);
                return SIM_FREE;
            }
        }
        polling = 0;
        onode* n = table_find(t, kb, kw, knp);
        if (n && n->expanding[pl]) return SIM_WAIT_EXPAND; /* while key in self.now_expanding: await asyncio.sleep(...) */
        if (!(n && n->expanded[pl])) {
            if (n && !c->share_mtcs_info) g->n_mirror_hits++; /* reached a key only mirror writes created */
            expand_begin(g, pl, s);
            return SIM_WAIT_NET;
        }
        if (s->depth >= 64) return -1;
        const double vl = (double)c->virtual_loss;
        const double vlw = knp == 1 ? vl : -vl;
        const int a = select_action(g, pl, env, s->depth == 0);
        orc_env_step(env, a);
        n = table_get(t, kb, kw, knp);
        n->N[a] += vl;
        n->W[a] -= vlw;
        s->kb[s->depth] = kb; s->kw[s->depth] = kw; s->knp[s->depth] = knp; s->act[s->depth] = a;
        s->depth++;
    }
}

/* search_moves (:189-200) under raz-sched-v1: the asyncio event loop in EXACT VIRTUAL TIME - the
 * clock only moves when no coroutine is runnable, and then to the earliest timer; timers with equal
 * deadlines fire in the order they were set.  (The reference's real interleaving at
 * parallel_search_num > 1 depends on how long Python takes between its 100 us / 10 us sleeps and is
 * not reproducible run to run; raz-sched-v1 is that loop with computation taking no time.  The
 * UNMODIFIED reference player run on such a loop - oracle/ref_harness.py VirtualTimeLoop - produces
 * the goldens this function is pinned to, tests/golden/mcts_par_games.json.)  What the loop then
 * does between two wake-ups of prediction_worker (every prediction_worker_sleep_sec = 10 ticks of
 * wait_for_expanding_sleep_sec) is one round:
 *   B  the worker drained the queue into one api.predict call and resolved the futures: every
 *      simulation waiting for the net finishes its expansion and returns up its path, in the order
 *      the leaves were queued (:343-349); each return releases the semaphore, waking the next
 *      waiting simulation BEHIND the remaining resumptions;
 *   C  the woken simulations run in turn, each until it blocks (a simulation that ends on a
 *      finished game releases its slot to the next one at once);
 *   D  one tick later the simulations sleeping on now_expanding poll (:253-254) in the order they
 *      went to sleep: those whose key has been expanded in B go on from where they stood, the
 *      others (blocked by a leaf queued in C) sleep on; slots released here are refilled after
 *      the whole batch of sleepers has polled (C').
 * The first round of a search has an empty B: the worker's first look at the queue comes right
 * after the initial `parallel_search_num` simulations have blocked. */
typedef struct {
    osim* sims;
    osim* queue[ORC_MAX_PAR];    /* prediction_queue, put order */
    osim* sleepers[ORC_MAX_PAR]; /* simulations sleeping on now_expanding, poll order */
    int nq, ns, K, to_start, inflight;
} osched;

static int sched_fill(ogame* g, int pl, osched* sc, u64 own, u64 enemy, int* sims_done) {
    while (sc->inflight < sc->K && sc->to_start > 0) {
        int j = 0;
        while (sc->sims[j].state != SIM_FREE) ++j;
        osim* s = &sc->sims[j];
        s->depth = 0;
        orc_env_update(&s->env, own, enemy, 1); /* ReversiEnv().update(own, enemy, Player.black) (:209) */
        --sc->to_start;
        const int r = descend(g, pl, s, 0);
        if (r < 0) return -1;
        s->state = r;
        if (r == SIM_FREE) { g->n_sims++; ++*sims_done; continue; }
        ++sc->inflight;
        if (r == SIM_WAIT_NET) sc->queue[sc->nq++] = s;
        else { sc->sleepers[sc->ns++] = s; g->n_parked++; }
    }
    return 0;
}

static int search_moves(ogame* g, int pl, u64 own, u64 enemy, int sims_per_move, int* sims_done) {
    osched sc;
    memset(&sc, 0, sizeof sc);
    sc.K = g->cfg->parallel_search_num;
    sc.sims = (osim*)calloc((size_t)sc.K, sizeof(osim));
    sc.to_start = sims_per_move;
    int rc = 0;
    while (rc == 0 && (sc.to_start > 0 || sc.inflight > 0)) {
        /* B */
        osim* batch[ORC_MAX_PAR];
        const int nb = sc.nq;
        memcpy(batch, sc.queue, sizeof batch);
        sc.nq = 0;
        for (int i = 0; i < nb; ++i) {
            expand_finish(g, pl, batch[i]);
            batch[i]->state = SIM_FREE; --sc.inflight; g->n_sims++; ++*sims_done;
        }
        /* C */
        if ((rc = sched_fill(g, pl, &sc, own, enemy, sims_done)) != 0) break;
        /* D: every sleeper polls once, in order; one that sleeps on keeps its place */
        osim* poll[ORC_MAX_PAR];
        const int np = sc.ns;
        memcpy(poll, sc.sleepers, sizeof poll);
        sc.ns = 0;
        for (int i = 0; i < np && rc == 0; ++i) {
            const int r = descend(g, pl, poll[i], 1);
            if (r < 0) { rc = -1; break; }
            poll[i]->state = r;
            if (r == SIM_WAIT_EXPAND) sc.sleepers[sc.ns++] = poll[i];
            else if (r == SIM_WAIT_NET) sc.queue[sc.nq++] = poll[i];
            else { --sc.inflight; g->n_sims++; ++*sims_done; }
        }
        /* C' */
        if (rc == 0) rc = sched_fill(g, pl, &sc, own, enemy, sims_done);
    }
    free(sc.sims);
    return rc;
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
            n->has_p = 1;
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
            if (search_moves(g, pl, own, enemy, sims_per_move, &rec->sims) != 0) return -2; /* :189-215 */
        } else { /* bypass_first_move (:143-148) */
            onode* n = table_get(t, root.black, root.white, 1);
            u64 legal = orc_find_correct_moves(root.black, root.white);
            int first = 0, cnt = orc_bit_count(legal);
            while (!((legal >> first) & 1)) ++first;
            n->N[first] = 1;
            n->W[first] = 0;
            for (int i = 0; i < 64; ++i) n->P[i] = (float)((double)((legal >> i) & 1) / (double)cnt);
            n->has_p = 1;
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

/* MCTSInfo carried from game to game by one worker (worker/self_play.py:109-111: created when None
 * and share_mtcs_info_in_self_play; :132-134: dropped every reset_mtcs_info_per_game games). */
struct orc_tree {
    otable table;
};
orc_tree* orc_tree_new(void) {
    orc_tree* t = (orc_tree*)calloc(1, sizeof *t);
    table_init(&t->table);
    return t;
}
void orc_tree_free(orc_tree* t) {
    if (!t) return;
    table_free(&t->table);
    free(t);
}

/* SelfPlayWorker.start_game (worker/self_play.py:139-175) for one game; plies[] needs room for
 * max_plies records.  Returns the number of plies recorded, or -1 on error.
 * tree (nullable): the worker's MCTSInfo; used - and left holding this game's statistics too - when
 * share_mtcs_info.  The game's two new ReversiPlayers start with expanded = set(var_p.keys())
 * (agent/player.py:47) and an empty now_expanding. */
static int selfplay_impl(orc_tree* tree, const orc_play_cfg* cfg, const void* blob, size_t blob_bytes, orc_nn_fn nn,
                         void* nn_ctx, uint32_t seed, uint32_t game_id, int sims_per_move, orc_ply_record* plies,
                         int max_plies, int stop_after_plies, orc_game_summary* sum, const orc_env* start) {
    ogame g;
    memset(&g, 0, sizeof g);
    g.cfg = cfg; g.blob = blob; g.blob_bytes = blob_bytes; g.seed = seed; g.game_id = game_id;
    g.nn = nn; g.nn_ctx = nn_ctx;
    if (!nn && !blob) return -1;
    if (cfg->parallel_search_num < 1 || cfg->parallel_search_num > ORC_MAX_PAR) return -1;
    const int carried = tree && cfg->share_mtcs_info;
    g.solver[0] = orc_solver_new();
    g.solver[1] = orc_solver_new();
    if (carried) {
        g.tables[0] = tree->table;
        for (size_t i = 0; i < g.tables[0].cap; ++i) {
            onode* n = &g.tables[0].nodes[i];
            if (!n->used) continue;
            n->expanded[0] = n->expanded[1] = n->has_p;
            n->expanding[0] = n->expanding[1] = 0;
        }
    } else {
        table_init(&g.tables[0]);
    }
    if (!cfg->share_mtcs_info) table_init(&g.tables[1]);
    double d[2];
    orc_rng_pair(seed, game_id, 3, 0, 0, 0, d);
    int enable_resign = cfg->disable_resignation_rate <= d[0]; /* self_play.py:144 */
    orc_env env;
    if (start) env = *start;   /* a game taken up at a position (ReversiEnv.update, env/reversi_env.py:33-40): fresh players, fresh streams */
    else orc_env_reset(&env);
    int np = 0;
    while (!env.done) {
        if (stop_after_plies > 0 && np >= stop_after_plies) break; /* partial replay (spot checks): the prefix is the game's */
        if (np >= max_plies) { np = -1; break; }
        int pl = env.next_player == 1 ? 0 : 1;
        u64 own = pl == 0 ? env.black : env.white, enemy = pl == 0 ? env.white : env.black;
        int a = action_with_evaluation(&g, pl, own, enemy, sims_per_move, enable_resign, &plies[np]);
        if (a == -2) { np = -1; break; }
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
        sum->n_parked = g.n_parked;
    }
    orc_solver_free(g.solver[0]);
    orc_solver_free(g.solver[1]);
    if (carried) tree->table = g.tables[0]; /* (the table may have been re-allocated while growing) */
    else table_free(&g.tables[0]);
    if (!cfg->share_mtcs_info) table_free(&g.tables[1]);
    return np;
}

int orc_selfplay_game_ex(orc_tree* tree, const orc_play_cfg* cfg, const void* blob, size_t blob_bytes, orc_nn_fn nn,
                         void* nn_ctx, uint32_t seed, uint32_t game_id, int sims_per_move, orc_ply_record* plies,
                         int max_plies, int stop_after_plies, orc_game_summary* sum) {
    return selfplay_impl(tree, cfg, blob, blob_bytes, nn, nn_ctx, seed, game_id, sims_per_move, plies, max_plies, stop_after_plies, sum, NULL);
}

/* The same loop taken up at (black, white, next_player) instead of the initial position: what SelfPlayWorker.start_game
 * would do on an env put there by ReversiEnv.update (env/reversi_env.py:33-40) - two new players, empty trees, the game's
 * random streams at their first events.  Used to check games the engine continues from a given position
 * (raz_engine_set_position with one_move = 0: bench.py's steady-state batch). */
int orc_selfplay_game_from(const orc_play_cfg* cfg, const void* blob, size_t blob_bytes, orc_nn_fn nn, void* nn_ctx,
                           uint32_t seed, uint32_t game_id, int sims_per_move, u64 black, u64 white, int next_player,
                           orc_ply_record* plies, int max_plies, int stop_after_plies, orc_game_summary* sum) {
    orc_env start;
    orc_env_reset(&start);
    orc_env_update(&start, black, white, next_player);
    return selfplay_impl(NULL, cfg, blob, blob_bytes, nn, nn_ctx, seed, game_id, sims_per_move, plies, max_plies, stop_after_plies, sum, &start);
}

int orc_selfplay_game_on(orc_tree* tree, const orc_play_cfg* cfg, const void* blob, size_t blob_bytes, uint32_t seed,
                         uint32_t game_id, int sims_per_move, orc_ply_record* plies, int max_plies,
                         orc_game_summary* sum) {
    return orc_selfplay_game_ex(tree, cfg, blob, blob_bytes, NULL, NULL, seed, game_id, sims_per_move, plies, max_plies, 0, sum);
}

int orc_selfplay_game(const orc_play_cfg* cfg, const void* blob, size_t blob_bytes, uint32_t seed,
                      uint32_t game_id, int sims_per_move, orc_ply_record* plies, int max_plies,
                      orc_game_summary* sum) {
    return orc_selfplay_game_on(NULL, cfg, blob, blob_bytes, seed, game_id, sims_per_move, plies, max_plies, sum);
}
