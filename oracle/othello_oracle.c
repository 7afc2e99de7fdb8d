/*
 * othello_oracle.c - restatement of the reference Othello bitboard (src/cpp/Othello.h) and
 * the Othello instantiation of the oracle's PUCT search.  TEST INFRASTRUCTURE ONLY.
 *
 * Bit layout (Othello.h:18-27): bit i = row i/8, col i%8; bb[0] = Black (+1), bb[1] = White
 * (-1); actions 0-63 = squares, 64 = pass.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define NOT_A_FILE 0xFEFEFEFEFEFEFEFEULL
#define NOT_H_FILE 0x7F7F7F7F7F7F7F7FULL

static int popc(uint64_t x) { return __builtin_popcountll(x); }

/* Othello.h:133-148: 0=N 1=NE 2=E 3=SE 4=S 5=SW 6=W 7=NW with wrap-around masks */
static uint64_t shift_dir(uint64_t b, int d)
{
    switch (d) {
    case 0: return b >> 8;
    case 1: return (b >> 7) & NOT_A_FILE;
    case 2: return (b << 1) & NOT_A_FILE;
    case 3: return (b << 9) & NOT_A_FILE;
    case 4: return b << 8;
    case 5: return (b << 7) & NOT_H_FILE;
    case 6: return (b >> 1) & NOT_H_FILE;
    case 7: return (b >> 9) & NOT_H_FILE;
    default: return 0;
    }
}

/* Othello.h:63-78 */
void oro_ot_reset(oro_ot *s)
{
    memset(s->cells, 0, sizeof s->cells);
    s->turn = 1;
    s->cells[3 * 8 + 3] = -1; s->cells[3 * 8 + 4] = 1;
    s->cells[4 * 8 + 3] = 1;  s->cells[4 * 8 + 4] = -1;
    s->bb[0] = (1ULL << 28) | (1ULL << 35);
    s->bb[1] = (1ULL << 27) | (1ULL << 36);
    s->n_pieces = 4;
    s->passes = 0;
    s->last_player = -1;
}

/* Othello.h:87-111: import forgets any pass made before this position */
void oro_ot_import(oro_ot *s, const int8_t *cells64)
{
    memcpy(s->cells, cells64, ORO_OT_CELLS);
    s->bb[0] = s->bb[1] = 0;
    for (int i = 0; i < ORO_OT_CELLS; ++i) {
        if (s->cells[i] == 1) s->bb[0] |= 1ULL << i;
        else if (s->cells[i] == -1) s->bb[1] |= 1ULL << i;
    }
    s->n_pieces = popc(s->bb[0]) + popc(s->bb[1]);
    s->passes = 0;
    s->last_player = -1;
}

/* Othello.h:114-125 */
void oro_ot_export_cells(oro_ot *s)
{
    memset(s->cells, 0, sizeof s->cells);
    for (int i = 0; i < 64; ++i) {
        if (s->bb[0] & (1ULL << i)) s->cells[i] = 1;
        else if (s->bb[1] & (1ULL << i)) s->cells[i] = -1;
    }
}

/* Othello.h:155-171 */
uint64_t oro_ot_valid_positions(const oro_ot *s)
{
    int p = (s->turn == 1) ? 0 : 1;
    uint64_t own = s->bb[p], opp = s->bb[1 - p], empty = ~(own | opp), valid = 0;
    for (int d = 0; d < 8; ++d) {
        uint64_t c = shift_dir(own, d) & opp;
        for (int i = 0; i < 5; ++i) c |= shift_dir(c, d) & opp;
        valid |= shift_dir(c, d) & empty;
    }
    return valid;
}

/* Othello.h:177-198 */
static uint64_t compute_flips(const oro_ot *s, int pos)
{
    int p = (s->turn == 1) ? 0 : 1;
    uint64_t own = s->bb[p], opp = s->bb[1 - p], placed = 1ULL << pos, flipped = 0;
    for (int d = 0; d < 8; ++d) {
        uint64_t cand = 0, sq = shift_dir(placed, d);
        while (sq & opp) { cand |= sq; sq = shift_dir(sq, d); }
        if (sq & own) flipped |= cand;
    }
    return flipped;
}

/* Othello.h:206-235 */
void oro_ot_step(oro_ot *s, int action)
{
    if (action == ORO_OT_PASS) { s->passes++; s->turn = -s->turn; return; }
    int p = (s->turn == 1) ? 0 : 1;
    uint64_t placed = 1ULL << action, flips = compute_flips(s, action);
    s->bb[p] |= placed | flips;
    s->bb[1 - p] &= ~flips;
    s->cells[action] = (int8_t)s->turn;
    for (uint64_t f = flips; f; f &= f - 1) s->cells[__builtin_ctzll(f)] = (int8_t)s->turn;
    s->n_pieces++;
    s->passes = 0;
    s->last_player = p;
    s->turn = -s->turn;
}

/* Othello.h:241-244, 296-302 */
int oro_ot_full(const oro_ot *s) { return s->n_pieces == 64 || s->passes >= 2; }

/* Othello.h:250-258 */
int oro_ot_winner(const oro_ot *s)
{
    if (!oro_ot_full(s)) return 0;
    int a = popc(s->bb[0]), b = popc(s->bb[1]);
    return a > b ? 1 : (b > a ? -1 : 0);
}

/* Othello.h:283-294: ascending squares, or the single pass action */
int oro_ot_valid_moves(const oro_ot *s, int *moves65)
{
    if (oro_ot_full(s)) return 0;
    uint64_t v = oro_ot_valid_positions(s);
    if (v == 0) { moves65[0] = ORO_OT_PASS; return 1; }
    int n = 0;
    for (; v; v &= v - 1) moves65[n++] = __builtin_ctzll(v);
    return n;
}

/* Othello.h:312-326 */
void oro_ot_transform_coord(int sym, int r, int c, int *nr, int *nc)
{
    switch (sym) {
    case 1: *nr = c;     *nc = 7 - r; break;
    case 2: *nr = 7 - r; *nc = 7 - c; break;
    case 3: *nr = 7 - c; *nc = r;     break;
    case 4: *nr = r;     *nc = 7 - c; break;
    case 5: *nr = 7 - r; *nc = c;     break;
    case 6: *nr = c;     *nc = r;     break;
    case 7: *nr = 7 - c; *nc = 7 - r; break;
    default: *nr = r;    *nc = c;     break;
    }
}

static uint64_t transform_bb(uint64_t b, int sym) /* Othello.h:329-341 */
{
    if (sym == 0) return b;
    uint64_t r = 0;
    for (uint64_t bits = b; bits; bits &= bits - 1) {
        int i = __builtin_ctzll(bits), nr, nc;
        oro_ot_transform_coord(sym, i / 8, i % 8, &nr, &nc);
        r |= 1ULL << (nr * 8 + nc);
    }
    return r;
}

void oro_ot_apply_sym(oro_ot *s, int sym) /* Othello.h:347-353 */
{
    if (sym == 0) return;
    s->bb[0] = transform_bb(s->bb[0], sym);
    s->bb[1] = transform_bb(s->bb[1], sym);
    oro_ot_export_cells(s);
}

static int inverse_sym(int sym) /* Othello.h:356-361 */
{
    static const int inv[8] = {0, 3, 2, 1, 4, 5, 6, 7};
    return inv[sym];
}

/* Othello.h:373-387 */
static void ot_inverse_sym_policy(const float *src, int sym, float *dst)
{
    if (sym == 0) { memcpy(dst, src, sizeof(float) * ORO_OT_ACTIONS); return; }
    int inv = inverse_sym(sym);
    for (int i = 0; i < 64; ++i) {
        int nr, nc;
        oro_ot_transform_coord(inv, i / 8, i % 8, &nr, &nc);
        dst[nr * 8 + nc] = src[i];
    }
    dst[ORO_OT_PASS] = src[ORO_OT_PASS];
}

/* Othello.h:363-367: index into {0, 2, 6, 7} */
static int ot_sample_sym(orc_mt19937 *g)
{
    static const int ids[4] = {0, 2, 6, 7};
    return ids[orc_uniform_int(g, 0, 3)];
}

/* Othello.h:260-266 */
static float ot_terminal_aux(const oro_ot *s, const orc_config *cfg)
{
    int diff = popc(s->bb[0]) - popc(s->bb[1]);
    float raw = (float)(diff * s->turn);
    return atanf(raw / cfg->score_scale) * (2.0f / 3.14159265f);
}

/* Othello.h:268-274 (child_M arrives negated, MCTS.h:197-198) */
static float ot_aux_utility(float child_M, float parent_M, float child_Q, const orc_config *cfg)
{
    (void)parent_M; (void)child_Q;
    if (cfg->score_utility_factor <= 0.0f) return 0.0f;
    return cfg->score_utility_factor * child_M;
}

/* env_common.h:93-119 */
void oro_ot_current_state(oro_ot *s, float *out192)
{
    oro_ot_export_cells(s);
    memset(out192, 0, sizeof(float) * 3 * ORO_OT_CELLS);
    for (int i = 0; i < ORO_OT_CELLS; ++i) {
        int8_t v = s->cells[i];
        if (v == s->turn) out192[i] = 1.0f;
        else if (v == -s->turn) out192[ORO_OT_CELLS + i] = 1.0f;
        out192[2 * ORO_OT_CELLS + i] = (float)s->turn;
    }
}

/* ------------------------------------------------------------------ search instantiation */
#define PFX(name) oro_##name
#define G_BATCH oro_batch
#define G_BATCH_T oro_batch
#define G_ACTIONS ORO_OT_ACTIONS
#define G_CELLS ORO_OT_CELLS
#define G_MAX_PATH 160 /* <= 60 placements plus passes */
#define g_state oro_ot
#define g_reset oro_ot_reset
#define g_import oro_ot_import
#define g_step oro_ot_step
#define g_winner oro_ot_winner
#define g_full oro_ot_full
#define g_valid_moves oro_ot_valid_moves
#define g_apply_sym oro_ot_apply_sym
#define g_sample_sym(rng) ot_sample_sym(rng)
#define g_inverse_sym_policy ot_inverse_sym_policy
#define G_AUX_PLUS_ONE_PER_PLY 0
#define G_AUX_NEGATE_PER_PLY 1
#define g_aux_utility ot_aux_utility
#define g_terminal_aux(state, cfg) ot_terminal_aux((state), (cfg))

#include "mcts_impl.inc"
