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

/* Connect4.h:288-294: the mirror is its own inverse */
static void c4_inverse_sym_policy(const float *src, int sym, float *dst)
{
    for (int a = 0; a < ORC_C4_ACTIONS; ++a) dst[a] = src[sym ? (ORC_C4_ACTIONS - 1 - a) : a];
}

#define PFX(name) orc_##name
#define G_BATCH orc_batch
#define G_BATCH_T orc_batch
#define G_ACTIONS ORC_C4_ACTIONS
#define G_CELLS ORC_C4_CELLS
#define G_MAX_PATH 64 /* a Connect4 descent has at most 42 plies */
#define g_state orc_c4
#define g_reset orc_c4_reset
#define g_import orc_c4_import
#define g_step orc_c4_step
#define g_winner orc_c4_winner
#define g_full orc_c4_full
#define g_valid_moves orc_c4_valid_moves
#define g_apply_sym orc_c4_mirror
#define g_sample_sym(rng) orc_uniform_int((rng), 0, 1) /* NUM_SYMMETRIES = 2 */
#define g_inverse_sym_policy c4_inverse_sym_policy
#define G_AUX_PLUS_ONE_PER_PLY 1
#define G_AUX_NEGATE_PER_PLY 0
#define g_aux_utility c4_aux_utility
#define g_terminal_aux(state, cfg) 0.0f /* Connect4.h:226-229 */

#include "mcts_impl.inc"
