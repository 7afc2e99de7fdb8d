/*
 * oracle.h - CPU restatement of the AlphaZero-AL batched MCTS hot path (Connect4).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (alphazero-al_amd/, bench.py's timed
 * GPU path) may include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * What it restates (paths relative to the reference checkout):
 *   src/cpp/Connect4.h      bitboard game                     -> c4_oracle.c
 *   src/cpp/MCTSNode.h      WDL / node / edge / pool / config -> mcts_oracle.c
 *   src/cpp/MCTS.h          single-tree PUCT + virtual loss   -> mcts_oracle.c
 *   src/cpp/BatchedMCTS.h   per-env batch layer (serial here) -> mcts_oracle.c
 *   src/cpp/RolloutEvaluator.h random playout evaluator       -> mcts_oracle.c
 *   src/cpp/Othello.h       bitboard game + its search        -> othello_oracle.c
 *   (mcts_impl.inc is the game-generic body both searches are compiled from)
 *   libstdc++-11 <random>   mt19937 / uniform_int / gamma     -> rng_oracle.c
 *
 * Parity pin: checked bit-for-bit against the real reference compiled into oracle/_ref
 * (tests/golden/make_golden.py writes the fixtures; tests/test_oracle_golden.py replays
 * them).  The reference has no tests or golden vectors of its own.
 *
 * Floating point: the oracle is compiled with -ffp-contract=off and writes the three
 * fused multiply-adds that g++ -O3 on an FMA machine puts into the reference binary
 * (setup.py:24-27 uses -march=native) explicitly with fmaf(); see mcts_oracle.c.
 *
 * RNG model: ONE mt19937 stream consumed in env order - i.e. the reference run with
 * OMP_NUM_THREADS=1 (BatchedMCTS.h:68-84 seeds thread t with seed + t*10007).
 */
#ifndef AZ_ORACLE_H
#define AZ_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- rng_oracle.c */

typedef struct {
    uint32_t mt[624];
    int      idx;
} orc_mt19937;

/* std::normal_distribution<float> keeps its second polar variate (random.tcc:1802-1836) */
typedef struct {
    float saved;
    int   saved_available;
} orc_normal_state;

void     orc_mt_seed(orc_mt19937 *g, uint32_t seed);
uint32_t orc_mt_next(orc_mt19937 *g);
/* std::uniform_int_distribution<int>(a, b)(g): Lemire, bits/uniform_int_dist.h:243-271 */
int      orc_uniform_int(orc_mt19937 *g, int a, int b);
/* std::generate_canonical<float, 24>(g), random.tcc:3348-3385 */
float    orc_canonical_float(orc_mt19937 *g);
/* std::normal_distribution<float>(0, 1)(g) with its cached second variate */
float    orc_normal_float(orc_mt19937 *g, orc_normal_state *st);
/* `count` draws from ONE fresh std::gamma_distribution<float>(alpha, 1) object */
void     orc_gamma_fill(orc_mt19937 *g, float alpha, float *out, int count);

/* ---------------------------------------------------------------- c4_oracle.c */

enum { ORC_C4_ROWS = 6, ORC_C4_COLS = 7, ORC_C4_CELLS = 42, ORC_C4_ACTIONS = 7 };

typedef struct {
    int8_t   cells[ORC_C4_CELLS]; /* display grid, row 0 = top (Connect4.h:51)        */
    int      turn;                /* side to move, +1 / -1 (Connect4.h:52)            */
    uint64_t bb[2];               /* bb[0] = player +1, bb[1] = player -1             */
    int      height[ORC_C4_COLS]; /* next free bit index of each column               */
    int      n_pieces;
    int      last_player;         /* index of the last mover, -1 if none              */
} orc_c4;

void orc_c4_reset(orc_c4 *s);
void orc_c4_import(orc_c4 *s, const int8_t *cells42);
void orc_c4_export_cells(orc_c4 *s);               /* sync_to_board                    */
void orc_c4_step(orc_c4 *s, int col);
int  orc_c4_winner(const orc_c4 *s);               /* +1 / -1 / 0                      */
int  orc_c4_full(const orc_c4 *s);
int  orc_c4_valid_moves(const orc_c4 *s, int *moves7); /* ascending; returns count     */
void orc_c4_mirror(orc_c4 *s, int sym_id);         /* apply_symmetry                   */
/* env_common.h:93-119 make_current_state -> float[3*42] */
void orc_c4_current_state(orc_c4 *s, float *out126);

/* ---------------------------------------------------------------- othello_oracle.c */

enum { ORO_OT_CELLS = 64, ORO_OT_ACTIONS = 65, ORO_OT_PASS = 64 };

typedef struct {
    int8_t   cells[ORO_OT_CELLS]; /* display grid (Othello.h:51)                       */
    int      turn;
    uint64_t bb[2];               /* bit i = row i/8, col i%8                          */
    int      n_pieces;
    int      passes;              /* consecutive passes since import (Othello.h:56)    */
    int      last_player;
} oro_ot;

void     oro_ot_reset(oro_ot *s);
void     oro_ot_import(oro_ot *s, const int8_t *cells64);
void     oro_ot_export_cells(oro_ot *s);
void     oro_ot_step(oro_ot *s, int action);
int      oro_ot_winner(const oro_ot *s);
int      oro_ot_full(const oro_ot *s);            /* == is_game_over                    */
uint64_t oro_ot_valid_positions(const oro_ot *s);
int      oro_ot_valid_moves(const oro_ot *s, int *moves65);
void     oro_ot_apply_sym(oro_ot *s, int sym_id);
void     oro_ot_transform_coord(int sym, int r, int c, int *nr, int *nc);
void     oro_ot_current_state(oro_ot *s, float *out192);

/* ---------------------------------------------------------------- mcts_oracle.c */

typedef struct {
    float c_init;               /* MCTSNode.h:49-60 defaults                            */
    float c_base;
    float dirichlet_alpha;
    float noise_epsilon;
    float fpu_reduction;
    float mlh_slope;
    float mlh_cap;
    float score_utility_factor;
    float score_scale;
    float value_decay;
    int   use_symmetry;
    int   vl_count;
} orc_config;

typedef struct orc_batch orc_batch;

/* running totals since creation / orc_stats_reset (engine-independent workload figures) */
typedef struct {
    int64_t sims;          /* simulate / simulate_vl calls                             */
    int64_t levels;        /* select_edge calls (descent levels)                       */
    int64_t expansions;    /* expand_leaf calls                                        */
    int64_t terminal_hits; /* sims that ended on a terminal leaf                       */
    int64_t dup_leaves;    /* backprop_vl on a leaf an earlier k of the call expanded  */
    int64_t backup_nodes;  /* nodes updated by propagate                               */
} orc_stats;

orc_batch  *orc_create(int n_envs);
void        orc_destroy(orc_batch *b);
orc_config *orc_config_ptr(orc_batch *b);
int         orc_num_envs(const orc_batch *b);
void        orc_set_seed(orc_batch *b, int seed);
void        orc_reset_env(orc_batch *b, int env);
void        orc_prune_roots(orc_batch *b, const int32_t *actions);

void orc_search_batch(orc_batch *b, const int8_t *boards, const int32_t *turns,
                      int8_t *out_boards, float *out_d, float *out_p1w, float *out_p2w,
                      uint8_t *out_is_term, int32_t *out_turns, uint8_t *out_valid_mask);
void orc_backprop_batch(orc_batch *b, const float *policy, const float *d, const float *p1w,
                        const float *p2w, const float *moves_left, const uint8_t *is_term);
void orc_remove_all_vl(orc_batch *b, int K);
void orc_search_batch_vl(orc_batch *b, int K, const int8_t *boards, const int32_t *turns,
                         int8_t *out_boards, float *out_d, float *out_p1w, float *out_p2w,
                         uint8_t *out_is_term, int32_t *out_turns, int32_t *out_sym_ids,
                         uint8_t *out_valid_mask);
void orc_backprop_batch_vl(orc_batch *b, int K, const float *policy, const float *d,
                           const float *p1w, const float *p2w, const float *moves_left,
                           const uint8_t *is_term, const int32_t *sym_ids);
/* BatchedMCTS::search with RolloutEvaluator (BatchedMCTS.h:339-407) */
void orc_search_rollout(orc_batch *b, const int8_t *boards, const int32_t *turns, int n_playout);

void orc_get_all_counts(const orc_batch *b, int32_t *out /* n_envs*7 */);
void orc_get_all_root_stats(const orc_batch *b, float *out /* n_envs*62 */);

/* oracle-only: the symmetry ids of the last search_batch (pending_sym_ids_, BatchedMCTS.h:45) */
void orc_pending_sym(const orc_batch *b, int32_t *out /* n_envs */);
void orc_stats_get(const orc_batch *b, orc_stats *out);
void orc_stats_reset(orc_batch *b);
/* pool occupancy of one tree: nodes / edges in use */
void orc_tree_size(const orc_batch *b, int env, int32_t *nodes, int32_t *edges);

/* The same search instantiated for Othello (othello_oracle.c): identical signatures, boards
 * of 64 cells, 65 actions, 526 statistics per tree. */
typedef struct oro_batch oro_batch;
oro_batch  *oro_create(int n_envs);
void        oro_destroy(oro_batch *b);
orc_config *oro_config_ptr(oro_batch *b);
int         oro_num_envs(const oro_batch *b);
void        oro_set_seed(oro_batch *b, int seed);
void        oro_reset_env(oro_batch *b, int env);
void        oro_prune_roots(oro_batch *b, const int32_t *actions);
void oro_search_batch(oro_batch *b, const int8_t *boards, const int32_t *turns,
                      int8_t *out_boards, float *out_d, float *out_p1w, float *out_p2w,
                      uint8_t *out_is_term, int32_t *out_turns, uint8_t *out_valid_mask);
void oro_backprop_batch(oro_batch *b, const float *policy, const float *d, const float *p1w,
                        const float *p2w, const float *moves_left, const uint8_t *is_term);
void oro_remove_all_vl(oro_batch *b, int K);
void oro_search_batch_vl(oro_batch *b, int K, const int8_t *boards, const int32_t *turns,
                         int8_t *out_boards, float *out_d, float *out_p1w, float *out_p2w,
                         uint8_t *out_is_term, int32_t *out_turns, int32_t *out_sym_ids,
                         uint8_t *out_valid_mask);
void oro_backprop_batch_vl(oro_batch *b, int K, const float *policy, const float *d,
                           const float *p1w, const float *p2w, const float *moves_left,
                           const uint8_t *is_term, const int32_t *sym_ids);
void oro_search_rollout(oro_batch *b, const int8_t *boards, const int32_t *turns, int n_playout);
void oro_get_all_counts(const oro_batch *b, int32_t *out /* n_envs*65 */);
void oro_get_all_root_stats(const oro_batch *b, float *out /* n_envs*526 */);
void oro_pending_sym(const oro_batch *b, int32_t *out /* n_envs */);
void oro_stats_get(const oro_batch *b, orc_stats *out);
void oro_stats_reset(oro_batch *b);
void oro_tree_size(const oro_batch *b, int env, int32_t *nodes, int32_t *edges);

#ifdef __cplusplus
}
#endif
#endif
