// kernels.h - launch interface between the host engine (engine.hip) and the gfx950 kernels
// (kernels.hip).  Everything here is plain data: device pointers and sizes.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "tree_layout.h"

namespace az {

// Search parameters of one launch: a by-value snapshot of the live az_search_config
// (MCTSNode.h:47-61) plus what the device needs besides.
struct SearchParams {
    float c_init, c_base, noise_eps, fpu_reduction, mlh_slope, mlh_cap, value_decay, alpha;
    float score_utility_factor;
    int   vl_count;
    int   use_symmetry;
    // c_puct(parent_n) = c_init + logf((parent_n + c_base + 1)/c_base) for parent_n < tab_n,
    // tabulated by the HOST libm so that the device takes the same branch the reference's
    // logf takes (MCTS.h:213-214).  cpuct_tab[tab_n + parent_n] = sqrtf(parent_n) (exact on both sides; a load
    // instead of the refinement sequence).
    const float *cpuct_tab;
    int   tab_n;
    // Othello terminal_aux = atanf(diff*turn / score_scale) * (2/pi) (Othello.h:260-266) for
    // diff*turn in [-64, 64], tabulated by the host libm for the same reason
    const float *term_aux_tab;
    // device generator key (used by the dev_* entry points only).  `call_ptr` points at a
    // counter in HBM that a one-thread kernel bumps after every use, so that a captured
    // hipGraph draws fresh numbers on every replay.
    uint64_t seed;
    const uint64_t *call_ptr;
    // per-tree root-noise epsilon (n trees, device memory) replacing `noise_eps` when set: the
    // self-play driver's noise decay over the plies of a game (game.py:87-91), where games have ages
    const float *noise_eps_tree;
};

struct TreeArena {
    HotRec  *hot;    // [B][2][S]: two halves per tree (tree_layout.h)
    ColdRec *cold;
    uint8_t *half;   // [B] the half a tree lives in now
    int32_t *root;   // [B] slot of each tree's root, relative to its half
    int32_t *used;   // [B] slots in use in that half
    int64_t  S;      // slots per half
    int      B;      // trees
};

// Root positions of the current call: two bitboards + side to move + the game's small integer
// (Connect4: index of the last mover, -1 on an empty board, Connect4.h:124-128; Othello:
// consecutive passes, 0 after an import, Othello.h:108-110)
struct RootState {
    uint64_t *bb0, *bb1;
    int32_t  *turn;
    int32_t  *aux;
};

// What a descent leaves behind for expansion/backup; flat index = tree*K + k
struct LeafBuf {
    int32_t  *slot;      // leaf node
    uint64_t *bb0, *bb1; // leaf position (unsymmetrised)
    int32_t  *turn;      // side to move at the leaf
    int32_t  *aux;       // the game's small integer at the leaf (see RootState)
    uint8_t  *nvalid;    // legal moves at the leaf (what an expansion will append)
    uint8_t  *flags;     // LEAF_*
    int32_t  *path_len;  // nodes on the path, root first (0 = no descent recorded)
    int32_t  *path;      // [flat*MAX_PATH + depth]
    int32_t  *sym;       // symmetry id shown to the evaluator
};

// Evaluator outputs for one backprop launch
struct EvalIn {
    const float   *policy;    // [n*K, A] in the frame the evaluator saw
    // host form (reference signature): absolute WDL and flags supplied by the caller
    const float   *d, *p1w, *p2w;
    const uint8_t *is_term;
    const int32_t *sym;       // VL host form: per leaf; nullptr -> LeafBuf::sym
    // fused form: relative [draw, win, loss] per leaf, terminal flag from LeafBuf
    const float   *wdl_rel;
    const float   *moves_left;
    // Dirichlet noise for root expansions, [B, A], already normalised (host RNG mode);
    // nullptr -> device generator
    const float   *root_noise;
};

// Device transposition table (tt_kernels.hip): entries of tt_entry_bytes(game) bytes - key (two u64 XORed
// with a checksum of the value), policy[A], relative wdl[3], auxiliary value, a stamp (0 = never written),
// padded to 64-byte lines: Connect4 64 B, Othello 320 B - in buckets of four
struct TtTable {
    void *e;
    uint64_t mask;                                  // entries - 1 (entries = power of two >= 4)
    unsigned long long *stats;                      // lookups, hits, inserts, replaced
};
size_t tt_entry_bytes(int game);

// bits of the engine's sticky device error word
constexpr int ERR_ARENA_OVERFLOW = 1;   // an expansion found no room in its tree's arena (it was dropped)
constexpr int ERR_LIST_OVERFLOW  = 2;   // a compact leaf list was asked to hold more entries than leaves exist

enum : int { CNT_SIMS = 0, CNT_LEVELS, CNT_EXPANSIONS, CNT_TERMINAL, CNT_DUP, CNT_BACKUP,
             CNT_SELECT_LAUNCHES, CNT_BACKPROP_LAUNCHES, CNT_N };
constexpr int CNT_STRIPES = 64;      // striped copies of the counters (CNT_N * 8 B = one 64-byte line each)
static_assert(CNT_N == 8, "one counter stripe is one 64-byte line");

// Every launcher takes the game id (AZ_GAME_*) and dispatches to the kernel instantiation.
void launch_import(int game, const int8_t *boards, const int32_t *turns, RootState rs, int B, hipStream_t s);
void launch_set_roots(int game, const uint64_t *bb0, const uint64_t *bb1, const int32_t *turns, RootState rs,
                      int B, hipStream_t s);
void launch_bump_call(uint64_t *call_ctr, hipStream_t s);
// bump_call: the device generator's call counter, incremented once by the launch (nullptr: not)
// returns the name of the kernel it launched (a string literal)
const char *launch_select(int game, TreeArena ar, RootState rs, LeafBuf lf, SearchParams p, int K, bool vl,
                          unsigned long long *counters, hipStream_t s, uint64_t *bump_call = nullptr,
                          int64_t *zero = nullptr);   // zero: an int64 the launch clears (the live-leaf count)
void launch_backprop(int game, TreeArena ar, LeafBuf lf, SearchParams p, int K, bool vl, bool fused,
                     EvalIn in, unsigned long long *counters, int *err, hipStream_t s);
void launch_remove_vl(int game, TreeArena ar, LeafBuf lf, SearchParams p, int K, int strideK, hipStream_t s);
// Leaves -> evaluator input.  gen_sym: draw symmetry ids from the device generator (else read
// LeafBuf::sym).  Any output pointer may be nullptr.
void launch_export(int game, LeafBuf lf, SearchParams p, int n_leaves, bool gen_sym, int8_t *boards,
                   uint8_t *valid_mask, float *features, hipStream_t s);
// dev_noise: fresh root noise is written by the launch itself - from the device generator, or from
// `replay_noise` ([B, A] by edge index) when that is given
// A tree that occupies more than `compact_above` records has its kept subtree moved into its other half
// (compaction), the others re-root in place; max_live: device int that receives, by atomic max, the number
// of records a tree occupies afterwards; err: the engine's error word.
void launch_prune(int game, TreeArena ar, SearchParams p, const int32_t *actions, int32_t *noise_req,
                  bool dev_noise, hipStream_t s, const float *replay_noise, int *max_live, int *err, int compact_above);
void launch_apply_noise(int game, TreeArena ar, const int32_t *noise_req, const float *noise, hipStream_t s);
void launch_reset_masked(TreeArena ar, const uint8_t *mask, hipStream_t s);
void launch_counts(int game, TreeArena ar, int32_t *counts, hipStream_t s);
void launch_root_stats(int game, TreeArena ar, float *stats, hipStream_t s);
void launch_init_trees(TreeArena ar, hipStream_t s);
void launch_rollout(int game, LeafBuf lf, SearchParams p, int B, float *policy, float *d, float *p1w, float *p2w,
                    float *ml, uint8_t *is_term, hipStream_t s);
void launch_game_step(int game, uint64_t *bb0, uint64_t *bb1, int32_t *turns, int32_t *aux, const int32_t *actions,
                      uint8_t *done, int32_t *winner, int64_t n, bool reset_finished, hipStream_t s);
void launch_game_valid_mask(int game, const uint64_t *bb0, const uint64_t *bb1, const int32_t *turns, const int32_t *aux,
                            uint8_t *mask, int64_t n, hipStream_t s);

// err: the engine's sticky error word (ERR_* bits)
// symmetry ids + action masks + (idx != nullptr) the compact list of non-terminal leaves, for evaluators that read
// leaf positions; `count` must have been cleared (the selection launch does)
void launch_leaf_prep(int game, LeafBuf lf, SearchParams p, int n_leaves, bool gen_sym, uint8_t *valid_mask, int32_t *idx,
                      int64_t *count, int *err, hipStream_t s);
void launch_live_leaves(LeafBuf lf, int n_leaves, int32_t *idx, int64_t *count, int *err, hipStream_t s, bool clear_count = true);
void launch_tt_lookup(int game, LeafBuf lf, int n_leaves, TtTable t, const uint64_t *clock, float *probs, float *wdl, float *ml,
                      int32_t *miss_idx, int64_t *miss_count, uint64_t *keys, int *err, hipStream_t s);
void launch_tt_insert(int game, int n_leaves, TtTable t, const uint64_t *clock, const int32_t *miss_idx, const int64_t *miss_count,
                      const uint64_t *keys, const float *probs, const float *wdl, const float *ml, hipStream_t s);

// table refresh (MCTS_cpp.py:361-377): entries [e0, e0+n) -> positions + masks + compact list; fresh values -> same keys
void launch_tt_refresh_gather(int game, TtTable t, uint64_t e0, int n, uint64_t *bb_p1, uint64_t *bb_p2, int32_t *turn, int32_t *sym,
                              uint8_t *mask, int32_t *rows, int64_t *count, uint64_t *keys, hipStream_t s);
void launch_tt_refresh_store(int game, TtTable t, uint64_t e0, int n, const int32_t *rows, const int64_t *count, const uint64_t *keys,
                             const float *probs, const float *wdl, const float *ml, hipStream_t s);

}  // namespace az
