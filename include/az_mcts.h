/*
 * az_mcts.h - C ABI of the MI355X batched self-play search engine (libaz_mcts.so).
 *
 * This is the drop-in boundary for the reference's native search layer: every entry point
 * below replaces one method the reference binds in src/cpp/mcts_bindings.cpp (cited per
 * function, paths relative to the reference checkout).  Plain pointers and sizes only; no
 * torch / pybind types.  All trees live in HBM; the library fails (non-zero return,
 * message in az_last_error()) when no HIP device is usable - there is no CPU fallback.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error (AZ_ERR_*); the reference
 *     throws std::runtime_error at the same points (mcts_bindings.cpp:76-79,97-100,
 *     155-165,206-212,278-288,323-327) and the pybind layer re-raises RuntimeError.
 *   - "host" entry points take host pointers, are synchronous, and mirror the reference
 *     signature 1:1 (caller-allocated outputs instead of fresh numpy arrays).
 *   - "dev" entry points take DEVICE pointers and a hipStream_t (as void*), never
 *     synchronise, and are what the fused self-play loop uses (no host round trip).
 *   - flat leaf index of simulation k of tree i in a K-wide call is i*K + k
 *     (BatchedMCTS.h:221,251).
 */
#ifndef AZ_MCTS_H
#define AZ_MCTS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZ_OK            0
#define AZ_ERR_ARG       1   /* size / shape / range mismatch (reference: runtime_error) */
#define AZ_ERR_DEVICE    2   /* HIP error or no device                                   */
#define AZ_ERR_CAPACITY  3   /* a tree arena overflowed (cannot happen unless growth is disabled) */
#define AZ_ERR_STATE     4   /* call sequence not supported                              */

#define AZ_GAME_CONNECT4 0
#define AZ_GAME_OTHELLO  1

/* Same fields, order and defaults as the reference SearchConfig (MCTSNode.h:47-61),
 * exposed as a LIVE struct exactly like `BatchedMCTS.config` (mcts_bindings.cpp:55-58):
 * the engine re-reads it at every call. */
typedef struct az_search_config {
    float   c_init;               /* 1.25    */
    float   c_base;               /* 19652   */
    float   dirichlet_alpha;      /* 0.3     */
    float   noise_epsilon;        /* 0.25    */
    float   fpu_reduction;        /* 0.4     */
    float   mlh_slope;            /* 0       */
    float   mlh_cap;              /* 0.2     */
    float   score_utility_factor; /* 0       */
    float   score_scale;          /* 8       */
    float   value_decay;          /* 1       */
    uint8_t use_symmetry;         /* true    */
    int32_t vl_count;             /* 1       */
} az_search_config;

typedef struct az_mcts az_mcts;

/* thread-local message of the last failing call */
const char *az_last_error(void);

/* static game geometry: BatchedMCTS_<G>.action_size / board_size (mcts_bindings.cpp:359-369) */
int az_game_action_size(int game);
int az_game_board_size(int game);
int az_game_board_rows(int game);
int az_game_board_cols(int game);

/* ---- lifetime ------------------------------------------------------------------------ */

/* BatchedMCTS(int n_envs), mcts_bindings.cpp:52 / BatchedMCTS.h:52-58.  device < 0 uses
 * the current HIP device. */
int  az_mcts_create(int game, int n_envs, int device, az_mcts **out);
void az_mcts_destroy(az_mcts *m);
/* `.config` property, mcts_bindings.cpp:55-58 */
az_search_config *az_mcts_config(az_mcts *m);
/* get_num_envs, mcts_bindings.cpp:68 */
int  az_mcts_num_envs(const az_mcts *m);
/* set_seed, mcts_bindings.cpp:61 / BatchedMCTS.h:68-84.  Seeds the host mt19937 that feeds
 * symmetry ids and Dirichlet noise in "reference RNG" mode (== the reference run with
 * OMP_NUM_THREADS=1) and the counter-based device generator used by the dev entry points. */
int  az_mcts_set_seed(az_mcts *m, int seed);
/* reset_env, mcts_bindings.cpp:65 / BatchedMCTS.h:93-99 (out-of-range index is ignored) */
int  az_mcts_reset_env(az_mcts *m, int env);
/* prune_roots, mcts_bindings.cpp:72-81 / MCTS.h:90-132; n must equal n_envs */
int  az_mcts_prune_roots(az_mcts *m, const int32_t *actions, int64_t n);

/* ---- host entry points (reference signatures) ----------------------------------------- */

/* search_batch, mcts_bindings.cpp:89-134 / BatchedMCTS.h:119-171 */
int az_mcts_search_batch(az_mcts *m, const int8_t *boards, const int32_t *turns, int64_t n,
                         int8_t *out_boards, float *out_term_d, float *out_term_p1w,
                         float *out_term_p2w, uint8_t *out_is_term, int32_t *out_turns,
                         uint8_t *out_valid_mask);
/* backprop_batch, mcts_bindings.cpp:139-179 / BatchedMCTS.h:176-199 */
int az_mcts_backprop_batch(az_mcts *m, const float *policy, const float *d, const float *p1w,
                           const float *p2w, const float *moves_left, const uint8_t *is_term,
                           int64_t n);
/* remove_all_vl, mcts_bindings.cpp:184-191 / BatchedMCTS.h:209-216 */
int az_mcts_remove_all_vl(az_mcts *m, int K);
/* search_batch_vl, mcts_bindings.cpp:197-252 / BatchedMCTS.h:227-286 */
int az_mcts_search_batch_vl(az_mcts *m, int K, const int8_t *boards, const int32_t *turns,
                            int64_t n, int8_t *out_boards, float *out_term_d,
                            float *out_term_p1w, float *out_term_p2w, uint8_t *out_is_term,
                            int32_t *out_turns, int32_t *out_sym_ids, uint8_t *out_valid_mask);
/* backprop_batch_vl, mcts_bindings.cpp:257-306 / BatchedMCTS.h:296-332; total must be n_envs*K */
int az_mcts_backprop_batch_vl(az_mcts *m, int K, const float *policy, const float *d,
                              const float *p1w, const float *p2w, const float *moves_left,
                              const uint8_t *is_term, const int32_t *sym_ids, int64_t total);
/* search(RolloutEvaluator, ...), mcts_bindings.cpp:313-337 / BatchedMCTS.h:339-407 with
 * RolloutEvaluator.h:23-48, in the reference's random stream like every host entry point: playout
 * moves (one uniform_int per move, leaves in env order) and root-noise rows come from the host
 * mt19937 as the reference with OMP_NUM_THREADS=1 consumes them - bit-exact against it (fixture G9);
 * selection, expansion and backup run on the device, one host round trip per playout. */
int az_mcts_search_rollout(az_mcts *m, const int8_t *boards, const int32_t *turns, int64_t n,
                           int n_playout);
/* The same search with the random playouts on the device as well (moves and noise from the device
 * generator: same distribution, another stream; no host round trip inside the loop). */
int az_mcts_search_rollout_dev(az_mcts *m, const int8_t *boards, const int32_t *turns, int64_t n,
                               int n_playout);
/* get_all_counts, mcts_bindings.cpp:342 / BatchedMCTS.h:413-427: out[n_envs*A] */
int az_mcts_get_all_counts(az_mcts *m, int32_t *out);
/* get_all_root_stats, mcts_bindings.cpp:348-356 / MCTS.h:637-673: out[n_envs*(6+8A)] */
int az_mcts_get_all_root_stats(az_mcts *m, float *out);

/* ---- device entry points (fused loop; no reference equivalent - they are what removes the
 *      four host crossings per iteration of MCTS_cpp.py:250-357) --------------------------- */

/* Must be called OUTSIDE any stream capture before the other dev_* calls and again whenever
 * the live config changed: sizes the leaf buffers for K descents per tree, refreshes the
 * c_puct table, and guarantees arena room for `sims_per_tree` more simulations per tree
 * (grows the arenas if needed; synchronises).  Reserve PER SEARCH: the host-side occupancy bound restarts
 * from the figure each az_mcts_dev_prune_roots reports, so room reserved before a re-rooting for
 * searches enqueued after it is forgotten - call this once for every search, after the re-rooting that
 * precedes it (as selfplay.py / fused.py do).  The room is sims_per_tree x the game's largest block of children
 * (Connect4 7; Othello 33, the most legal moves of a position REACHABLE in play): imported Othello positions with
 * more moves are searched correctly but can outrun the reservation - an expansion that does not fit is dropped and
 * raises the sticky error word (az_mcts_dev_check: arena full), never a store out of range. */
int az_mcts_dev_prepare(az_mcts *m, int K, int64_t sims_per_tree);
/* The same for a caller whose work on these trees is all on ONE stream: when the call has to look
 * at the trees (their fill), move a buffer or refresh a table, it waits for that stream only
 * instead of the whole device - what a driver with several engines on several streams wants. */
int az_mcts_dev_prepare_stream(az_mcts *m, int K, int64_t sims_per_tree, void *stream);
/* Sticky device error word, polled without stalling: enqueues a copy of the word to pinned host
 * memory on `stream` and reports what the PREVIOUS poll brought back - AZ_ERR_CAPACITY (message
 * in az_last_error) once an expansion found its arena full or a compact leaf list overflowed.
 * A self-play driver calls it once per ply; az_mcts_counters reports the same word synchronously. */
int az_mcts_dev_check(az_mcts *m, void *stream);
/* Test hook - recorded draws instead of the device generator.  The reference draws symmetry ids
 * and Dirichlet noise from one mt19937 in env order (BatchedMCTS.h:148-154,261-267; MCTS.h:113-132,
 * 352-358); the device loop has a counter-based generator of its own.  To compare the device loop
 * with the reference bit for bit, a test records what the reference (the oracle) drew and plays it
 * back: sym_ids int32 [n_select_calls][sym_stride] in DEVICE memory - the c-th az_mcts_dev_select
 * (or selection inside az_mcts_dev_search) after this call shows leaf `flat` under symmetry
 * sym_ids[c*sym_stride + flat] (0 for terminal leaves; running past the tape is AZ_ERR_STATE);
 * root_noise float [n_envs][A] in DEVICE memory, row = tree, column = EDGE index (legal moves in
 * ascending order), already normalised - what root expansions and az_mcts_dev_prune_roots store
 * from now on; the arrays are read when the kernels run, so update them in stream order.  Either
 * pointer may be NULL (that part stays with the generator); both NULL ends the replay. */
int az_mcts_dev_replay(az_mcts *m, const int32_t *sym_ids, int64_t sym_stride, int64_t n_select_calls,
                       const float *root_noise);
/* Root positions as bitboards already in HBM: bb_p1/bb_p2 uint64[n_envs], turn int32[n_envs]
 * (the reference passes int8 grids on every call, BatchedMCTS.h:136-137,244-245). */
int az_mcts_dev_set_roots(az_mcts *m, const uint64_t *bb_p1, const uint64_t *bb_p2,
                          const int32_t *turns, void *stream);
/* Same from int8 grids in HBM (device pointer), boards[n_envs*board_size]. */
int az_mcts_dev_import_roots(az_mcts *m, const int8_t *boards, const int32_t *turns, void *stream);
/* K descents per tree (K=1, vl=0: simulate; vl=1: simulate_vl), then gather of the leaves into
 * the evaluator's input: features float32[n_envs*K,3,rows,cols] in the reference's relative
 * planes (MCTS_cpp.py:15-20), action mask uint8[n_envs*K,A] (0 for terminal leaves).
 * Symmetry ids come from the device generator. */
int az_mcts_dev_select(az_mcts *m, int K, int vl, float *features, uint8_t *valid_mask,
                       void *stream);
/* Expansion + backup straight from the evaluator's outputs: probs float32[n*K,A] (leaf frame,
 * possibly mirrored), wdl_rel float32[n*K,3] = [draw, win, loss] for the side to move
 * (converted as MCTS_cpp.py:23-30), moves_left float32[n*K].  Terminal leaves ignore the
 * evaluator and use their cached result (MCTS_cpp.py:275-282). */
int az_mcts_dev_backprop(az_mcts *m, int K, int vl, const float *probs, const float *wdl_rel,
                         const float *moves_left, void *stream);
/* Leaf positions of the last dev_select (unsymmetrised): bb_p1/bb_p2 uint64[n*K],
 * turn int32[n*K], flags uint8[n*K] (bit 0 = terminal, bits 1-2 = result 0 draw / 1 P1 /
 * 2 P2).  Any pointer may be NULL. */
int az_mcts_dev_leaves(az_mcts *m, int K, uint64_t *bb_p1, uint64_t *bb_p2, int32_t *turns,
                       uint8_t *flags, void *stream);
/* Symmetry ids the leaves of the last dev_select (or of the last selection inside az_mcts_dev_search) were shown
 * under: int32[n*K] into HBM (0 for terminal leaves) - what BatchedMCTS.h:148-154,261-267 draws per leaf; lets a
 * test look at the device generator's draws themselves (tests/test_devrng_gpu.py). */
int az_mcts_dev_leaf_syms(az_mcts *m, int K, int32_t *sym_ids, void *stream);
/* Root visit counts int32[n_envs*A] / root stats float32[n_envs*(6+8A)] into HBM. */
int az_mcts_dev_counts(az_mcts *m, int32_t *counts, void *stream);
int az_mcts_dev_root_stats(az_mcts *m, float *stats, void *stream);
/* prune_roots with actions in HBM and Dirichlet noise from the device generator (or from az_mcts_dev_replay);
 * trees that run out of room are compacted into their other arena half on the way (see az_mcts_reserve). */
int az_mcts_dev_prune_roots(az_mcts *m, const int32_t *actions, void *stream);
/* Reset the trees whose mask byte is non-zero (mask uint8[n_envs] in HBM). */
int az_mcts_dev_reset_masked(az_mcts *m, const uint8_t *mask, void *stream);

/* Batched Connect4 positions in HBM - the device-side counterpart of Env.step / done /
 * winPlayer (env_common.h:141-147, env_connect4.h:38-40, Connect4.h:159-203) for a self-play
 * driver that keeps its games next to the trees: plays actions[i] in game i (bitboards and
 * side to move updated in place), writes done[i] (1 = won or board full) and winner[i]
 * (+1 / -1 / 0).  With reset_finished != 0 a finished game is replaced by the empty board
 * with player +1 to move.  Games with actions[i] < 0 are left untouched (done = 0). */
int az_c4_dev_step(uint64_t *bb_p1, uint64_t *bb_p2, int32_t *turns, const int32_t *actions,
                   uint8_t *done, int32_t *winner, int64_t n, int reset_finished, void *stream);
/* The same for either game (AZ_GAME_*).  `aux` [n] is the game's small integer carried from ply to
 * ply - Othello: consecutive passes so far (Othello.h:206-235, the Env keeps them although a tree
 * forgets them at every import, Othello.h:108-110); NULL: derived from the position as an import
 * derives it (enough for Connect4).  A finished game with reset_finished != 0 restarts from the
 * game's initial position (Connect4.h reset / Othello.h:62-75). */
int az_game_dev_step(int game, uint64_t *bb_p1, uint64_t *bb_p2, int32_t *turns, int32_t *aux,
                     const int32_t *actions, uint8_t *done, int32_t *winner, int64_t n, int reset_finished,
                     void *stream);
/* mask[i, a] = 1 iff action a is legal in position i (what `Env.valid_mask()` returns,
 * env_common.h; Othello: the pass action only when no placement exists, none after the end). */
int az_game_dev_valid_mask(int game, const uint64_t *bb_p1, const uint64_t *bb_p2, const int32_t *turns,
                           const int32_t *aux, uint8_t *mask, int64_t n, void *stream);

/* Root-noise epsilon PER TREE: `per_tree` (n_envs floats in DEVICE memory, owned and kept alive by the
 * caller; NULL switches back to the config's scalar) replaces az_search_config.noise_epsilon in every
 * selection from the next launch on.  The reference decays one global epsilon over the plies of a batch
 * of games that start together (game.py:87-91, AlphaZeroPlayer.noise_steps); a driver that refills
 * finished slots has games of all ages in one batch and needs the value per game.  The array is read by
 * the kernels when they run: update it in stream order. */
int az_mcts_dev_set_noise_epsilons(az_mcts *m, const float *per_tree);

/* The leaves of the last az_mcts_dev_select that an evaluator has to see - all but the terminal
 * ones, as the reference's wrapper evaluates them (MCTS_cpp.py:275-297): their flat indices go to
 * leaf_idx[0 .. *leaf_count) (int32 [n*K] and int64 [1] in DEVICE memory; order unspecified). */
int az_mcts_dev_live_leaves(az_mcts *m, int K, int32_t *leaf_idx, int64_t *leaf_count, void *stream);

/* A whole search as ONE call: the iteration schedule of the reference's wrapper
 * (MCTS_cpp.py:110-113, 217-264: one plain simulation that expands every root, then virtual-loss
 * batches of K until n_playout simulations per tree are done) with the evaluator inside the loop -
 * az_mcts_dev_select, az_mcts_dev_live_leaves (or the table's lookup / insert when use_table != 0),
 * az_nn_model_forward (include/az_nn.h) on the leaves that need it, az_mcts_dev_backprop - every
 * launch issued from native code on `stream`, the leaf batch in buffers the engine owns.  Roots
 * are the ones set by az_mcts_dev_set_roots / import_roots; results are read with
 * az_mcts_dev_counts / root_stats.  Identical to issuing the same calls one by one (tests).  Engines
 * on different streams (and host threads) overlap on the device: one engine's selection and
 * backup kernels, which leave most of the chip idle, run under another engine's evaluator. */
struct az_nn_model;
int az_mcts_dev_search(az_mcts *m, const struct az_nn_model *model, int n_playout, int K, int use_table,
                       void *stream);
/* CONTINUES a search on the same roots: n_sims more simulations per tree as whole iterations (virtual-loss batches of
 * min(K, remaining); plain simulations for K <= 1) WITHOUT the warm-up simulation az_mcts_dev_search starts with.  A
 * search under a time budget (MCTS_cpp.py:70-87,200-209,252-261: wall-clock check and top-2 early exit between
 * iterations) is az_mcts_dev_search(.., 1, ..) followed by chunks of this call, with az_mcts_dev_counts read once per
 * chunk - alphazero-al_amd/src/fused.py `search_timed`.  Every call leaves the trees without in-flight visits. */
int az_mcts_dev_search_more(az_mcts *m, const struct az_nn_model *model, int n_sims, int K, int use_table,
                            void *stream);

/* ---- device transposition table of evaluator outputs (both games) ------------------------
 * Replaces, for the device loop, the LRU table of the reference's wrapper (src/Cache.py:5-58 used
 * by src/MCTS_cpp.py:146-189 and 298-339): key = the symmetrised leaf position + side to move,
 * value = policy[A], relative wdl[3], auxiliary value.  2^log2_entries entries of 64 bytes (Connect4)
 * or 320 bytes (Othello), buckets of four, approximate-LRU replacement inside a bucket.  Between az_mcts_dev_select and
 * az_mcts_dev_backprop of one iteration:
 *   az_mcts_dev_tt_lookup   hits: the cached values are written to probs / wdl_rel / moves_left
 *                           at the leaf's flat index; misses: their flat indices are appended to
 *                           miss_idx[0 .. *miss_count) (int32 [n*K] and int64 [1], DEVICE memory;
 *                           the count is reset first).  Terminal leaves are neither.
 *   (evaluate the rows listed in miss_idx, writing the same three arrays - az_nn.h `batch_dev`)
 *   az_mcts_dev_tt_insert   stores the freshly evaluated rows.
 * A lookup never returns a value that was not inserted for exactly its key (torn entries fail a
 * checksum and read as misses).  When the evaluator's weights change the cached outputs are stale:
 * az_mcts_dev_tt_refresh re-evaluates them in place as the reference's `refresh_cache` does
 * (MCTS_cpp.py:361-377; both games), az_mcts_dev_tt_clear empties the table.
 * Statistics (synchronises): lookups, hits, inserts, entries replaced. */
int az_mcts_dev_tt_create(az_mcts *m, int log2_entries);
int az_mcts_dev_tt_clear(az_mcts *m, void *stream);
int az_mcts_dev_tt_lookup(az_mcts *m, int K, float *probs, float *wdl_rel, float *moves_left, int32_t *miss_idx,
                          int64_t *miss_count, void *stream);
int az_mcts_dev_tt_insert(az_mcts *m, int K, const int32_t *miss_idx, const int64_t *miss_count, const float *probs,
                          const float *wdl_rel, const float *moves_left, void *stream);
/* refresh_cache (MCTS_cpp.py:361-377): after a weight update every resident key is evaluated
 * again with `model` (include/az_nn.h) and keeps its place and age; entries that do not decode to
 * a position are emptied.  Enqueued on `stream` in chunks of 16384 entries. */
int az_mcts_dev_tt_refresh(az_mcts *m, const struct az_nn_model *model, void *stream);
int az_mcts_dev_tt_stats(az_mcts *m, int64_t out[4]);

/* ---- capacity / instrumentation ------------------------------------------------------- */

/* Tree arenas.  Every tree owns two halves of `capacity` node records and lives in one of them; a re-rooting
 * (prune_roots) copies the subtree it keeps into the other half when the tree could not take two more
 * searches' worth of growth where it is (the reference never reclaims a node before the next reset,
 * MCTS.h:90-108; node numbering is not observable).  az_mcts_reserve makes every half hold at least
 * `slots_per_tree` records (grows, never shrinks; synchronises); the engine grows by itself when a tree needs
 * more.  az_mcts_capacity: records per half. */
int az_mcts_reserve(az_mcts *m, int64_t slots_per_tree);
int64_t az_mcts_capacity(const az_mcts *m);
/* Changes whenever any device buffer the dev_* kernels address was reallocated (arena growth,
 * wider K, new c_puct table): a captured hipGraph of dev_* calls is valid for one epoch. */
int64_t az_mcts_epoch(const az_mcts *m);
/* Largest number of node records any tree uses right now (synchronises). */
int az_mcts_max_used(az_mcts *m, int64_t *out);

/* Workload counters since creation / last reset (synchronises):
 * [0] simulations [1] select levels [2] expansions [3] terminal leaves [4] duplicate VL
 * leaves [5] nodes updated by backup [6] select launches [7] backprop launches */
#define AZ_NUM_COUNTERS 8
int az_mcts_counters(az_mcts *m, int64_t out[AZ_NUM_COUNTERS]);
int az_mcts_counters_reset(az_mcts *m);

/* Kernel timing with HIP events recorded on the launch stream around every selection and every
 * expansion/backup kernel issued by the dev_* entry points (bench.py's roofline figures).
 * enable != 0 starts recording (at most AZ_PROFILE_MAX launches per kind are kept between
 * reads); enable = n > 1 times every n-th launch of a kind only (an event pair costs the stream
 * ~5 us, which a bench does not want around every launch); az_mcts_profile_read synchronises and returns, for [0] selection and [1]
 * expansion/backup, the summed kernel time in ms and the number of launches summed. */
#define AZ_PROFILE_MAX 8192
int az_mcts_profile(az_mcts *m, int enable);
int az_mcts_profile_read(az_mcts *m, double out_ms[2], int64_t out_launches[2]);
/* Name of the kernel behind the newest TIMED selection launch ("" before the first one): what the
 * engine launched, not what an environment knob asked for (static string, never NULL). */
const char *az_mcts_timed_select_kernel(az_mcts *m);

/* Test hook for the host generator: `count` Dirichlet-gamma draws from one fresh
 * gamma(alpha,1) object on an mt19937 seeded with `seed` (checked against libstdc++). */
int az_rng_gamma_selftest(uint32_t seed, float alpha, int count, float *out);

#ifdef __cplusplus
}
#endif
#endif
