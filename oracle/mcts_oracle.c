/*
 * mcts_oracle.c - restatement of the reference PUCT search for Connect4.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Follows (reference checkout):
 *   src/cpp/MCTSNode.h      13-39 WDLValue, 47-61 SearchConfig, 69-75 Edge, 85-140 node,
 *                           149-199 pool
 *   src/cpp/MCTS.h          77-132 reset/prune/noise, 140-234 fpu+select, 242-322 simulate,
 *                           329-413 expand/propagate/backprop, 421-609 virtual loss,
 *                           617-673 stats
 *   src/cpp/BatchedMCTS.h   68-84 seeding, 105-332 batch calls, 339-407 search(),
 *                           413-441 counts/stats
 *   src/cpp/RolloutEvaluator.h 23-48, src/cpp/Connect4.h 226-239, 288-294
 *
 * One tree per env, visited in env order on one thread: this is the reference with
 * OMP_NUM_THREADS=1, the only setting in which its seeded RNG streams are defined
 * independently of the machine.
 *
 * fp32 expression order is kept exactly; the three places where the compiled reference
 * uses a fused multiply-add (g++ -O3 -march=native/x86-64-v3, seen in its disassembly) are
 * written with fmaf(): FPU value, root prior/noise mix, value decay.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define A ORC_C4_ACTIONS
#define MAX_PATH 64 /* a Connect4 descent has at most 42 plies */

typedef struct { float d, p1w, p2w; } wdl_t;

typedef struct {            /* MCTSNode.h:69-75 */
    int32_t action, child;
    float   prior, noise;
} edge_t;

typedef struct {            /* MCTSNode.h:85-113 (sums, not means) */
    float   W_d, W_p1w, W_p2w;
    int32_t n_visits, n_inflight;
    float   M_sum;
    int32_t num_edges, edge_offset;
    int32_t parent, parent_edge;
    int8_t  turn;
    uint8_t expanded, terminal;
    wdl_t   term;
} node_t;

typedef struct { int32_t node, edge; } path_entry;

typedef struct {
    node_t *nodes; int node_cap, node_count;
    edge_t *edges; int edge_cap, edge_count;
    int32_t root;
    orc_c4  sim_env;
    int32_t leaf;          /* current_leaf_idx */
    /* virtual-loss state, MCTS.h:60-64 */
    int         vl_size;   /* == vl_paths_.size() */
    int         vl_cap;
    path_entry *paths;     /* vl_cap * MAX_PATH */
    int        *path_len;
    orc_c4     *vl_envs;
    int32_t    *vl_leaf;
} tree_t;

struct orc_batch {
    int         n;
    orc_config  cfg;
    tree_t     *trees;
    int        *pending_sym;   /* BatchedMCTS.h:45 */
    orc_mt19937 rng;           /* thread 0's engine */
    orc_stats   st;
};

/* ------------------------------------------------------------------ small helpers */

static wdl_t wdl_from_winner(int w) /* MCTSNode.h:35-39 */
{
    wdl_t r = {0, 0, 0};
    if (w == 1) r.p1w = 1.0f; else if (w == -1) r.p2w = 1.0f; else r.d = 1.0f;
    return r;
}

static float wdl_q(wdl_t w, int turn) { return (turn == 1) ? (w.p1w - w.p2w) : (w.p2w - w.p1w); }

static wdl_t node_mean_wdl(const node_t *n) /* MCTSNode.h:116-120 */
{
    wdl_t r;
    if (n->n_visits == 0) { r.d = r.p1w = r.p2w = 1.f / 3; return r; }
    float inv = 1.0f / (float)n->n_visits;
    r.d = n->W_d * inv; r.p1w = n->W_p1w * inv; r.p2w = n->W_p2w * inv;
    return r;
}

static float node_mean_q(const node_t *n) { return wdl_q(node_mean_wdl(n), n->turn); }

static float node_mean_M(const node_t *n) /* MCTSNode.h:131-133 */
{
    return (n->n_visits == 0) ? 0.0f : n->M_sum / (float)n->n_visits;
}

static wdl_t wdl_decayed(wdl_t w, float g) /* MCTSNode.h:28-31, contracted form */
{
    const float u = 1.0f / 3.0f;
    float c = (1 - g) * u;
    wdl_t r = {fmaf(w.d, g, c), fmaf(w.p1w, g, c), fmaf(w.p2w, g, c)};
    return r;
}

/* Connect4.h:231-239 */
static float c4_aux_utility(float child_M, float parent_M, float child_Q, const orc_config *cfg)
{
    if (cfg->mlh_slope <= 0.0f) return 0.0f;
    float m_diff = child_M - parent_M;
    float v = cfg->mlh_slope * m_diff;
    float lo = -cfg->mlh_cap, hi = cfg->mlh_cap;
    float m = (v < lo) ? lo : ((hi < v) ? hi : v); /* std::clamp */
    return m * child_Q;
}

/* ------------------------------------------------------------------ pool (MCTSNode.h:149-199) */

static int32_t alloc_node(tree_t *t)
{
    if (t->node_count >= t->node_cap) {
        t->node_cap *= 2;
        t->nodes = (node_t *)realloc(t->nodes, sizeof(node_t) * (size_t)t->node_cap);
    }
    int32_t idx = t->node_count++;
    node_t *n = &t->nodes[idx];
    memset(n, 0, sizeof *n);
    n->edge_offset = -1; n->parent = -1; n->parent_edge = -1; n->turn = 1;
    return idx;
}

static int32_t alloc_edges(tree_t *t, int count)
{
    int32_t off = t->edge_count;
    int need = t->edge_count + count;
    if (need > t->edge_cap) {
        int nc = t->edge_cap * 2;
        if (nc < need) nc = need;
        t->edge_cap = nc;
        t->edges = (edge_t *)realloc(t->edges, sizeof(edge_t) * (size_t)nc);
    }
    for (int i = 0; i < count; ++i) {
        edge_t *e = &t->edges[off + i];
        e->action = -1; e->child = -1; e->prior = 0.0f; e->noise = 0.0f;
    }
    t->edge_count = need;
    return off;
}

static void tree_reset(tree_t *t) /* MCTS.h:77-82 */
{
    t->node_count = 0;
    t->edge_count = 0;
    t->root = alloc_node(t);
    t->nodes[t->root].turn = 1;
}

static void tree_init(tree_t *t)
{
    memset(t, 0, sizeof *t);
    t->node_cap = 2048; t->edge_cap = 8192; /* MCTS.h:71, MCTSNode.h:152-158 */
    t->nodes = (node_t *)malloc(sizeof(node_t) * (size_t)t->node_cap);
    t->edges = (edge_t *)malloc(sizeof(edge_t) * (size_t)t->edge_cap);
    t->leaf = -1;
    orc_c4_reset(&t->sim_env);
    tree_reset(t);
}

static void tree_free(tree_t *t)
{
    free(t->nodes); free(t->edges); free(t->paths); free(t->path_len);
    free(t->vl_envs); free(t->vl_leaf);
}

/* ------------------------------------------------------------------ noise / prune */

/* MCTS.h:113-132 */
static void apply_root_noise(orc_batch *b, tree_t *t)
{
    if (b->cfg.dirichlet_alpha <= 0.0f) return;
    node_t *root = &t->nodes[t->root];
    if (!root->expanded || root->num_edges == 0) return;
    float g[A], sum = 0.0f;
    orc_gamma_fill(&b->rng, b->cfg.dirichlet_alpha, g, root->num_edges);
    for (int i = 0; i < root->num_edges; ++i) sum += g[i];
    float inv = 1.0f / (sum + 1e-8f);
    for (int i = 0; i < root->num_edges; ++i) t->edges[root->edge_offset + i].noise = g[i] * inv;
}

/* MCTS.h:90-108 */
static void prune_root(orc_batch *b, tree_t *t, int action)
{
    node_t *root = &t->nodes[t->root];
    if (root->expanded) {
        for (int i = 0; i < root->num_edges; ++i) {
            edge_t *e = &t->edges[root->edge_offset + i];
            if (e->action == action && e->child != -1) {
                t->root = e->child;
                t->nodes[t->root].parent = -1;
                apply_root_noise(b, t);
                return;
            }
        }
    }
    tree_reset(t);
}

/* ------------------------------------------------------------------ selection */

/* MCTS.h:140-156 */
static float compute_fpu(const orc_batch *b, const tree_t *t, int32_t idx)
{
    const node_t *node = &t->nodes[idx];
    float parent_q = node_mean_q(node);
    float seen = 0.0f;
    for (int i = 0; i < node->num_edges; ++i) {
        const edge_t *e = &t->edges[node->edge_offset + i];
        if (e->child != -1 && t->nodes[e->child].n_visits > 0) seen += e->prior;
    }
    float scale = (1.0f + parent_q) / 2.0f;
    float eff = b->cfg.fpu_reduction * scale;
    float fpu = fmaf(-eff, sqrtf(seen), parent_q); /* contracted pq - eff*sqrt(seen) */
    return (-1.0f < fpu) ? fpu : -1.0f;            /* std::max(-1.0f, fpu) */
}

/* MCTS.h:163-234 */
static int select_edge(const orc_batch *b, const tree_t *t, int32_t idx, float fpu)
{
    const orc_config *cfg = &b->cfg;
    const node_t *node = &t->nodes[idx];
    float parent_n = (float)(node->n_visits + node->n_inflight);
    float parent_M = node_mean_M(node);
    int is_root = (idx == t->root);
    float ne = cfg->noise_epsilon;

    float best_score = -INFINITY;
    int best = -1;
    for (int i = 0; i < node->num_edges; ++i) {
        const edge_t *e = &t->edges[node->edge_offset + i];
        float eff_prior = e->prior;
        if (is_root && ne > 0.0f)
            eff_prior = fmaf(e->prior, 1.0f - ne, ne * e->noise); /* contracted mix */

        float q, child_Q = 0.0f, child_M = 0.0f;
        int child_total = 0;
        int has_real = (e->child != -1 && t->nodes[e->child].n_visits > 0);
        if (has_real) {
            const node_t *c = &t->nodes[e->child];
            child_total = c->n_visits + c->n_inflight;
            child_Q = node_mean_q(c);
            child_M = node_mean_M(c);
            q = -child_Q;
        } else if (e->child != -1 && t->nodes[e->child].n_inflight > 0) {
            q = fpu;
            child_total = t->nodes[e->child].n_inflight;
        } else {
            q = fpu;
        }
        float c_puct = cfg->c_init + logf((parent_n + cfg->c_base + 1.0f) / cfg->c_base);
        float u = c_puct * eff_prior * sqrtf(parent_n) / (1.0f + (float)child_total);
        float m = has_real ? c4_aux_utility(child_M, parent_M, child_Q, cfg) : 0.0f;
        float score = q + u + m;
        if (score > best_score) { best_score = score; best = i; }
    }
    return best;
}

typedef struct { wdl_t wdl; int is_term; } sim_result;

/* MCTS.h:242-322 (vl_k < 0) and MCTS.h:443-545 (vl_k >= 0): one descent */
static sim_result descend(orc_batch *b, tree_t *t, const orc_c4 *start, int vl_k)
{
    b->st.sims++;
    t->sim_env = *start;
    int32_t cur = t->root;
    int winner = 0, full = 0, root_vl = 0;
    if (vl_k >= 0) t->path_len[vl_k] = 0;

    while (t->nodes[cur].expanded) {
        node_t *node = &t->nodes[cur];
        if (node->terminal) break;
        if (node->num_edges == 0) break;
        float fpu = compute_fpu(b, t, cur);
        int be = select_edge(b, t, cur, fpu);
        if (be < 0) break;
        b->st.levels++;

        if (vl_k >= 0 && !root_vl) { /* MCTS.h:470-475: AFTER the root's own selection */
            t->nodes[t->root].n_inflight += b->cfg.vl_count;
            root_vl = 1;
        }
        edge_t *e = &t->edges[node->edge_offset + be];
        orc_c4_step(&t->sim_env, e->action);
        if (e->child == -1) { /* lazy child, MCTS.h:268-275 */
            int32_t c = alloc_node(t);
            e = &t->edges[t->nodes[cur].edge_offset + be];
            e->child = c;
            t->nodes[c].parent = cur;
            t->nodes[c].parent_edge = be;
            t->nodes[c].turn = (int8_t)t->sim_env.turn;
        }
        if (vl_k >= 0) {
            t->nodes[e->child].n_inflight += b->cfg.vl_count;
            path_entry *p = &t->paths[vl_k * MAX_PATH + t->path_len[vl_k]++];
            p->node = cur; p->edge = be;
        }
        cur = e->child;

        winner = orc_c4_winner(&t->sim_env);
        full = orc_c4_full(&t->sim_env);
        if (winner != 0 || full) {
            t->nodes[cur].terminal = 1;
            t->nodes[cur].term = wdl_from_winner(winner);
            break;
        }
    }

    t->leaf = cur;
    if (vl_k >= 0) { t->vl_leaf[vl_k] = cur; t->vl_envs[vl_k] = t->sim_env; }

    sim_result r; r.wdl.d = r.wdl.p1w = r.wdl.p2w = 0.0f; r.is_term = 0;
    node_t *leaf = &t->nodes[cur];
    if (leaf->terminal) { r.wdl = leaf->term; r.is_term = 1; }
    else {
        if (winner == 0 && !full) { /* first visit of this state, MCTS.h:299-303 */
            winner = orc_c4_winner(&t->sim_env);
            full = orc_c4_full(&t->sim_env);
        }
        if (winner != 0 || full) {
            leaf->terminal = 1;
            leaf->term = wdl_from_winner(winner);
            r.wdl = leaf->term; r.is_term = 1;
        }
    }
    if (r.is_term) b->st.terminal_hits++;
    return r;
}

/* ------------------------------------------------------------------ expand / backup */

/* MCTS.h:329-375 */
static void expand_leaf(orc_batch *b, tree_t *t, const float *policy)
{
    int valids[A];
    int nv = orc_c4_valid_moves(&t->sim_env, valids);
    int32_t off = alloc_edges(t, nv);
    node_t *leaf = &t->nodes[t->leaf];
    leaf->edge_offset = off;
    leaf->num_edges = nv;
    leaf->expanded = 1;
    b->st.expansions++;

    float psum = 0.0f;
    for (int i = 0; i < nv; ++i) psum += policy[valids[i]];

    float noise[A] = {0};
    int has_noise = (leaf->parent == -1 && b->cfg.dirichlet_alpha > 0.0f);
    if (has_noise) {
        float sum = 0.0f;
        orc_gamma_fill(&b->rng, b->cfg.dirichlet_alpha, noise, nv);
        for (int i = 0; i < nv; ++i) sum += noise[i];
        float inv = 1.0f / (sum + 1e-8f);
        for (int i = 0; i < nv; ++i) noise[i] *= inv;
    }
    for (int i = 0; i < nv; ++i) {
        edge_t *e = &t->edges[off + i];
        e->action = valids[i];
        e->prior = policy[valids[i]] / (psum + 1e-8f);
        e->child = -1;
        if (has_noise) e->noise = noise[i];
    }
}

/* MCTS.h:381-402; Connect4 adds one ply to moves-left per level (Connect4.h:34) */
static void propagate(orc_batch *b, tree_t *t, wdl_t w, float ml)
{
    int32_t idx = t->leaf;
    while (idx != -1) {
        node_t *n = &t->nodes[idx];
        n->n_visits++;
        n->W_d += w.d; n->W_p1w += w.p1w; n->W_p2w += w.p2w;
        n->M_sum += ml;
        ml += 1.0f;
        idx = n->parent;
        b->st.backup_nodes++;
        if (b->cfg.value_decay < 1.0f) w = wdl_decayed(w, b->cfg.value_decay);
    }
}

/* MCTS.h:407-413 and 591-609; terminal_aux is 0 for Connect4 (Connect4.h:226-229) */
static void backprop_one(orc_batch *b, tree_t *t, const float *policy, wdl_t w, float ml,
                         int is_term, int vl)
{
    if (t->leaf == -1) return;
    if (!is_term) {
        if (!vl || !t->nodes[t->leaf].expanded) expand_leaf(b, t, policy);
        else b->st.dup_leaves++;
    }
    propagate(b, t, w, is_term ? 0.0f : ml);
}

/* ------------------------------------------------------------------ virtual loss bookkeeping */

/* MCTS.h:421-429 (std::vector::resize semantics) */
static void prepare_vl(tree_t *t, int K)
{
    if (K > t->vl_cap) {
        t->paths = (path_entry *)realloc(t->paths, sizeof(path_entry) * MAX_PATH * (size_t)K);
        t->path_len = (int *)realloc(t->path_len, sizeof(int) * (size_t)K);
        t->vl_envs = (orc_c4 *)realloc(t->vl_envs, sizeof(orc_c4) * (size_t)K);
        t->vl_leaf = (int32_t *)realloc(t->vl_leaf, sizeof(int32_t) * (size_t)K);
        t->vl_cap = K;
    }
    for (int k = t->vl_size; k < K; ++k) { t->vl_leaf[k] = -1; orc_c4_reset(&t->vl_envs[k]); }
    t->vl_size = K;
    for (int k = 0; k < K; ++k) t->path_len[k] = 0;
}

/* MCTS.h:561-581 */
static void remove_vl(orc_batch *b, tree_t *t, int K)
{
    int safe = (K < t->vl_size) ? K : t->vl_size;
    int vl = b->cfg.vl_count;
    for (int k = 0; k < safe; ++k) {
        if (t->path_len[k] > 0) t->nodes[t->root].n_inflight -= vl;
        for (int j = 0; j < t->path_len[k]; ++j) {
            path_entry *p = &t->paths[k * MAX_PATH + j];
            edge_t *e = &t->edges[t->nodes[p->node].edge_offset + p->edge];
            if (e->child != -1) t->nodes[e->child].n_inflight -= vl;
        }
        t->path_len[k] = 0;
    }
}

/* ------------------------------------------------------------------ batch layer */

orc_batch *orc_create(int n_envs)
{
    orc_batch *b = (orc_batch *)calloc(1, sizeof *b);
    b->n = n_envs;
    orc_config c = {1.25f, 19652.0f, 0.3f, 0.25f, 0.4f, 0.0f, 0.2f, 0.0f, 8.0f, 1.0f, 1, 1};
    b->cfg = c;
    b->trees = (tree_t *)malloc(sizeof(tree_t) * (size_t)n_envs);
    for (int i = 0; i < n_envs; ++i) tree_init(&b->trees[i]);
    b->pending_sym = (int *)calloc((size_t)n_envs, sizeof(int));
    orc_mt_seed(&b->rng, (uint32_t)time(NULL));
    return b;
}

void orc_destroy(orc_batch *b)
{
    if (!b) return;
    for (int i = 0; i < b->n; ++i) tree_free(&b->trees[i]);
    free(b->trees); free(b->pending_sym); free(b);
}

orc_config *orc_config_ptr(orc_batch *b) { return &b->cfg; }
int orc_num_envs(const orc_batch *b) { return b->n; }

/* BatchedMCTS.h:68-84, thread 0 */
void orc_set_seed(orc_batch *b, int seed)
{
    if (seed < 0) orc_mt_seed(&b->rng, (uint32_t)time(NULL) ^ (uint32_t)clock());
    else orc_mt_seed(&b->rng, (uint32_t)seed);
}

void orc_reset_env(orc_batch *b, int env) /* BatchedMCTS.h:93-99 */
{
    if (env >= 0 && env < b->n) tree_reset(&b->trees[env]);
}

void orc_prune_roots(orc_batch *b, const int32_t *actions) /* BatchedMCTS.h:105-112 */
{
    for (int i = 0; i < b->n; ++i) prune_root(b, &b->trees[i], actions[i]);
}

/* shared tail of search_batch / search_batch_vl (BatchedMCTS.h:141-169, 254-283) */
static void emit_leaf(orc_batch *b, tree_t *t, sim_result r, int flat, int8_t *out_boards,
                      float *out_d, float *out_p1w, float *out_p2w, uint8_t *out_is_term,
                      int32_t *out_turns, int32_t *sym_out, uint8_t *out_mask)
{
    orc_c4 board = t->sim_env;
    out_is_term[flat] = r.is_term ? 1 : 0;
    out_d[flat] = r.wdl.d; out_p1w[flat] = r.wdl.p1w; out_p2w[flat] = r.wdl.p2w;
    out_turns[flat] = board.turn;
    int sym = 0;
    if (!r.is_term && b->cfg.use_symmetry) {
        sym = orc_uniform_int(&b->rng, 0, 1); /* NUM_SYMMETRIES = 2 */
        if (sym != 0) orc_c4_mirror(&board, sym);
    }
    *sym_out = sym;
    memcpy(out_boards + (size_t)flat * ORC_C4_CELLS, board.cells, ORC_C4_CELLS);
    uint8_t *m = out_mask + (size_t)flat * A;
    memset(m, 0, A);
    if (!r.is_term) {
        int v[A], nv = orc_c4_valid_moves(&board, v);
        for (int i = 0; i < nv; ++i) m[v[i]] = 1;
    }
}

void orc_search_batch(orc_batch *b, const int8_t *boards, const int32_t *turns,
                      int8_t *out_boards, float *out_d, float *out_p1w, float *out_p2w,
                      uint8_t *out_is_term, int32_t *out_turns, uint8_t *out_valid_mask)
{
    for (int i = 0; i < b->n; ++i) {
        orc_c4 g;
        orc_c4_reset(&g);
        orc_c4_import(&g, boards + (size_t)i * ORC_C4_CELLS);
        g.turn = turns[i];
        sim_result r = descend(b, &b->trees[i], &g, -1);
        int32_t sym;
        emit_leaf(b, &b->trees[i], r, i, out_boards, out_d, out_p1w, out_p2w, out_is_term,
                  out_turns, &sym, out_valid_mask);
        b->pending_sym[i] = sym;
    }
}

static void unsym_policy(const float *src, int sym, float *dst) /* Connect4.h:288-294 */
{
    for (int a = 0; a < A; ++a) dst[a] = src[sym ? (A - 1 - a) : a];
}

void orc_backprop_batch(orc_batch *b, const float *policy, const float *d, const float *p1w,
                        const float *p2w, const float *moves_left, const uint8_t *is_term)
{
    for (int i = 0; i < b->n; ++i) {
        float pol[A];
        unsym_policy(policy + (size_t)i * A, b->pending_sym[i], pol);
        wdl_t w = {d[i], p1w[i], p2w[i]};
        backprop_one(b, &b->trees[i], pol, w, moves_left[i], is_term[i] != 0, 0);
    }
}

void orc_remove_all_vl(orc_batch *b, int K)
{
    for (int i = 0; i < b->n; ++i) remove_vl(b, &b->trees[i], K);
}

void orc_search_batch_vl(orc_batch *b, int K, const int8_t *boards, const int32_t *turns,
                         int8_t *out_boards, float *out_d, float *out_p1w, float *out_p2w,
                         uint8_t *out_is_term, int32_t *out_turns, int32_t *out_sym_ids,
                         uint8_t *out_valid_mask)
{
    for (int i = 0; i < b->n; ++i) {
        orc_c4 g;
        orc_c4_reset(&g);
        orc_c4_import(&g, boards + (size_t)i * ORC_C4_CELLS);
        g.turn = turns[i];
        tree_t *t = &b->trees[i];
        prepare_vl(t, K);
        for (int k = 0; k < K; ++k) {
            int flat = i * K + k;
            sim_result r = descend(b, t, &g, k);
            emit_leaf(b, t, r, flat, out_boards, out_d, out_p1w, out_p2w, out_is_term,
                      out_turns, &out_sym_ids[flat], out_valid_mask);
        }
    }
}

void orc_backprop_batch_vl(orc_batch *b, int K, const float *policy, const float *d,
                           const float *p1w, const float *p2w, const float *moves_left,
                           const uint8_t *is_term, const int32_t *sym_ids)
{
    for (int i = 0; i < b->n; ++i) {
        tree_t *t = &b->trees[i];
        remove_vl(b, t, K);
        for (int k = 0; k < K; ++k) {
            int flat = i * K + k;
            float pol[A];
            unsym_policy(policy + (size_t)flat * A, sym_ids[flat] != 0, pol);
            wdl_t w = {d[flat], p1w[flat], p2w[flat]};
            t->leaf = t->vl_leaf[k];           /* MCTS.h:595-597 */
            t->sim_env = t->vl_envs[k];
            backprop_one(b, t, pol, w, moves_left[flat], is_term[flat] != 0, 1);
        }
    }
}

/* BatchedMCTS.h:339-407 with RolloutEvaluator.h:23-48: uniform policy (all ones), random
 * playout to the end for the value, moves_left 0, no symmetry, no virtual loss. */
void orc_search_rollout(orc_batch *b, const int8_t *boards, const int32_t *turns, int n_playout)
{
    int n = b->n;
    sim_result *res = (sim_result *)malloc(sizeof(sim_result) * (size_t)n);
    wdl_t *ev = (wdl_t *)malloc(sizeof(wdl_t) * (size_t)n);
    for (int p = 0; p < n_playout; ++p) {
        for (int i = 0; i < n; ++i) {
            orc_c4 g;
            orc_c4_reset(&g);
            orc_c4_import(&g, boards + (size_t)i * ORC_C4_CELLS);
            g.turn = turns[i];
            res[i] = descend(b, &b->trees[i], &g, -1);
        }
        for (int i = 0; i < n; ++i) { /* evaluate_batch over the non-terminal leaves, in order */
            if (res[i].is_term) continue;
            orc_c4 sim = b->trees[i].sim_env;
            for (;;) {
                int w = orc_c4_winner(&sim);
                if (w != 0) { ev[i] = wdl_from_winner(w); break; }
                if (orc_c4_full(&sim)) { ev[i] = wdl_from_winner(0); break; }
                int v[A], nv = orc_c4_valid_moves(&sim, v);
                orc_c4_step(&sim, v[orc_uniform_int(&b->rng, 0, nv - 1)]);
            }
        }
        for (int i = 0; i < n; ++i) {
            float pol[A];
            for (int a = 0; a < A; ++a) pol[a] = res[i].is_term ? 0.0f : 1.0f;
            wdl_t w = res[i].is_term ? res[i].wdl : ev[i];
            backprop_one(b, &b->trees[i], pol, w, 0.0f, res[i].is_term, 0);
        }
    }
    free(res); free(ev);
}

/* ------------------------------------------------------------------ queries */

void orc_get_all_counts(const orc_batch *b, int32_t *out) /* MCTS.h:617-630 */
{
    memset(out, 0, sizeof(int32_t) * (size_t)b->n * A);
    for (int i = 0; i < b->n; ++i) {
        const tree_t *t = &b->trees[i];
        const node_t *root = &t->nodes[t->root];
        if (!root->expanded) continue;
        for (int j = 0; j < root->num_edges; ++j) {
            const edge_t *e = &t->edges[root->edge_offset + j];
            if (e->child != -1) out[i * A + e->action] = t->nodes[e->child].n_visits;
        }
    }
}

void orc_get_all_root_stats(const orc_batch *b, float *out) /* MCTS.h:637-673 */
{
    const int S = 6 + 8 * A;
    for (int i = 0; i < b->n; ++i) {
        const tree_t *t = &b->trees[i];
        const node_t *root = &t->nodes[t->root];
        float *o = out + (size_t)i * S;
        wdl_t rw = node_mean_wdl(root);
        o[0] = (float)root->n_visits; o[1] = node_mean_q(root); o[2] = node_mean_M(root);
        o[3] = rw.d; o[4] = rw.p1w; o[5] = rw.p2w;
        float *p = o + 6;
        memset(p, 0, sizeof(float) * 8 * A);
        if (!root->expanded) continue;
        for (int j = 0; j < root->num_edges; ++j) {
            const edge_t *e = &t->edges[root->edge_offset + j];
            float *slot = p + e->action * 8;
            slot[2] = e->prior; slot[3] = e->noise;
            if (e->child != -1) {
                const node_t *c = &t->nodes[e->child];
                wdl_t cw = node_mean_wdl(c);
                slot[0] = (float)c->n_visits; slot[1] = node_mean_q(c); slot[4] = node_mean_M(c);
                slot[5] = cw.d; slot[6] = cw.p1w; slot[7] = cw.p2w;
            }
        }
    }
}

void orc_stats_get(const orc_batch *b, orc_stats *out) { *out = b->st; }
void orc_stats_reset(orc_batch *b) { memset(&b->st, 0, sizeof b->st); }

void orc_tree_size(const orc_batch *b, int env, int32_t *nodes, int32_t *edges)
{
    *nodes = b->trees[env].node_count;
    *edges = b->trees[env].edge_count;
}
