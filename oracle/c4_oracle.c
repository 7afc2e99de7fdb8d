/*
 * c4_oracle.c - restatement of the reference Connect4 bitboard (src/cpp/Connect4.h).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Bit layout (Connect4.h:15-29): 7 bits per column (6 cells + 1 sentinel), bit index
 * col*7 + (5 - row) where row 0 is the TOP display row.
 */
#include "oracle.h"

#include <string.h>

#define BITS_PER_COL 7

static uint64_t col_mask(int c) { return 0x7FULL << (c * BITS_PER_COL); }

/* Connect4.h:62-72 */
void orc_c4_reset(orc_c4 *s)
{
    memset(s->cells, 0, sizeof s->cells);
    s->turn = 1;
    s->bb[0] = s->bb[1] = 0;
    s->n_pieces = 0;
    s->last_player = -1;
    for (int c = 0; c < ORC_C4_COLS; ++c) s->height[c] = c * BITS_PER_COL;
}

/* Connect4.h:87-129: copy the grid, rebuild bitboards bottom-up per column and stop at the
 * first empty cell; the last mover comes from piece-count parity, NOT from `turn`. */
void orc_c4_import(orc_c4 *s, const int8_t *cells42)
{
    memcpy(s->cells, cells42, ORC_C4_CELLS);
    s->bb[0] = s->bb[1] = 0;
    s->n_pieces = 0;
    s->last_player = -1;
    for (int c = 0; c < ORC_C4_COLS; ++c) {
        s->height[c] = c * BITS_PER_COL;
        for (int r = ORC_C4_ROWS - 1; r >= 0; --r) {
            int8_t v = s->cells[r * ORC_C4_COLS + c];
            if (v == 0) break;
            int p = (v == 1) ? 0 : 1;
            s->bb[p] |= 1ULL << s->height[c];
            s->height[c]++;
            s->n_pieces++;
        }
    }
    if (s->n_pieces > 0) s->last_player = (s->n_pieces % 2 == 1) ? 0 : 1;
}

/* Connect4.h:135-150 */
void orc_c4_export_cells(orc_c4 *s)
{
    memset(s->cells, 0, sizeof s->cells);
    for (int c = 0; c < ORC_C4_COLS; ++c) {
        int base = c * BITS_PER_COL;
        for (int bit = base; bit < s->height[c]; ++bit) {
            int row = ORC_C4_ROWS - 1 - (bit - base);
            s->cells[row * ORC_C4_COLS + c] = (s->bb[0] & (1ULL << bit)) ? 1 : -1;
        }
    }
}

/* Connect4.h:159-172 */
void orc_c4_step(orc_c4 *s, int col)
{
    int p = (s->turn == 1) ? 0 : 1;
    s->bb[p] |= 1ULL << s->height[col];
    int row = ORC_C4_ROWS - 1 - (s->height[col] - col * BITS_PER_COL);
    s->cells[row * ORC_C4_COLS + col] = (int8_t)s->turn;
    s->height[col]++;
    s->n_pieces++;
    s->last_player = p;
    s->turn = -s->turn;
}

/* Connect4.h:182-203: four-direction shift-AND on the LAST mover's bitboard */
int orc_c4_winner(const orc_c4 *s)
{
    if (s->last_player == -1) return 0;
    uint64_t b = s->bb[s->last_player];
    int result = (s->last_player == 0) ? 1 : -1;
    static const int dir[4] = {1, BITS_PER_COL, BITS_PER_COL - 1, BITS_PER_COL + 1};
    for (int i = 0; i < 4; ++i) {
        uint64_t t = b & (b >> dir[i]);
        if (t & (t >> (2 * dir[i]))) return result;
    }
    return 0;
}

/* Connect4.h:221-224 */
int orc_c4_full(const orc_c4 *s) { return s->n_pieces == ORC_C4_CELLS; }

/* Connect4.h:209-218: ascending column order - this is the edge order of a node */
int orc_c4_valid_moves(const orc_c4 *s, int *moves7)
{
    int n = 0;
    for (int c = 0; c < ORC_C4_COLS; ++c)
        if (s->height[c] < c * BITS_PER_COL + ORC_C4_ROWS) moves7[n++] = c;
    return n;
}

/* Connect4.h:249-280: sym 1 = swap columns c <-> 6-c in both bitboards and heights, then
 * rebuild the display grid from the bitboards. */
void orc_c4_mirror(orc_c4 *s, int sym_id)
{
    if (sym_id == 0) return;
    for (int p = 0; p < 2; ++p) {
        uint64_t src = s->bb[p], dst = src & col_mask(3);
        for (int c = 0; c < 3; ++c) {
            int sh = (6 - 2 * c) * BITS_PER_COL;
            dst |= (src & col_mask(c)) << sh;
            dst |= (src & col_mask(6 - c)) >> sh;
        }
        s->bb[p] = dst;
    }
    for (int c = 0; c < 3; ++c) {
        int m = ORC_C4_COLS - 1 - c;
        int hc = s->height[c] - c * BITS_PER_COL;
        int hm = s->height[m] - m * BITS_PER_COL;
        s->height[c] = c * BITS_PER_COL + hm;
        s->height[m] = m * BITS_PER_COL + hc;
    }
    orc_c4_export_cells(s);
}

/* env_common.h:93-119: planes [own, opponent, turn sign] from the display grid */
void orc_c4_current_state(orc_c4 *s, float *out126)
{
    orc_c4_export_cells(s);
    memset(out126, 0, sizeof(float) * 3 * ORC_C4_CELLS);
    for (int i = 0; i < ORC_C4_CELLS; ++i) {
        int8_t v = s->cells[i];
        if (v == s->turn) out126[i] = 1.0f;
        else if (v == -s->turn) out126[ORC_C4_CELLS + i] = 1.0f;
        out126[2 * ORC_C4_CELLS + i] = (float)s->turn;
    }
}
