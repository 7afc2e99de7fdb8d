// games.h - device-side game rules the search kernels are instantiated with.
//
// A game supplies: geometry, how many lanes cooperate on one tree (one lane per edge, so
// LANES >= the largest number of legal moves), the position update, the result test, legal
// moves in edge order, the evaluator-frame symmetry maps, and the game-specific auxiliary
// terms of the reference (`compute_aux_utility`, `terminal_aux`, per-ply change of the
// auxiliary value).  Positions are two u64 bitboards + side to move + one small integer.
// The rules are plain integer code, usable from host code as well (the reference-stream random
// playouts of az_mcts_search_rollout run on the host, where the reference's mt19937 lives).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.h"

namespace az {

struct GameState {
    uint64_t bb0, bb1;   // stones of player +1 / player -1
    int      turn;       // side to move
    int      aux;        // Connect4: index of the last mover (-1 none); Othello: consecutive passes
};

// ============================================================================ Connect4
// src/cpp/Connect4.h: 7 bits per column (6 cells + sentinel), bit = col*7 + (5 - row).
struct Connect4Dev {
    static constexpr int GAME_ID = 0;
    static constexpr int LANES = 8, ACTIONS = 7, ROWS = 6, COLS = 7, CELLS = 42;
    static constexpr int MAX_PATH = C4_MAX_PATH, STATS = 6 + 8 * ACTIONS;
    static constexpr int SYM_CHOICES = 2;                 // NUM_SYMMETRIES (Connect4.h:45)
    static constexpr bool AUX_PLUS_ONE = true, AUX_NEGATE = false;   // Connect4.h:34-35
    static constexpr int BPC = 7;

    __host__ __device__ static bool four(uint64_t b)               // Connect4.h:182-203
    {
        // all four directions, one test: on the device an early return per direction is a branch per direction
        uint64_t t, r;
        t = b & (b >> 1); r = t & (t >> 2);
        t = b & (b >> 7); r |= t & (t >> 14);
        t = b & (b >> 6); r |= t & (t >> 12);
        t = b & (b >> 8); r |= t & (t >> 16);
        return r != 0;
    }
    // last mover from piece parity, as import_board derives it (Connect4.h:124-128)
    __host__ __device__ static int root_aux(uint64_t bb0, uint64_t bb1)
    {
        const int pieces = __builtin_popcountll(bb0 | bb1);
        return pieces == 0 ? -1 : ((pieces & 1) ? 0 : 1);
    }
    __host__ __device__ static void start(GameState &s) { s.bb0 = 0; s.bb1 = 0; s.turn = 1; s.aux = -1; }
    __host__ __device__ static void import_cells(const int8_t *b, GameState &s)   // Connect4.h:87-129
    {
        uint64_t bb0 = 0, bb1 = 0;
        for (int c = 0; c < COLS; ++c) {
            int h = c * BPC;
            for (int r = ROWS - 1; r >= 0; --r) {
                const int8_t v = b[r * COLS + c];
                if (v == 0) break;
                if (v == 1) bb0 |= 1ull << h; else bb1 |= 1ull << h;
                ++h;
            }
        }
        s.bb0 = bb0; s.bb1 = bb1; s.aux = root_aux(bb0, bb1);
    }
    __host__ __device__ static void step(GameState &s, int action)  // Connect4.h:159-172
    {
        const uint64_t colmask = 0x7Full << (BPC * action);
        const uint64_t mv = (((s.bb0 | s.bb1) & colmask) + (1ull << (BPC * action))) & colmask;
        const int mover = (s.turn == 1) ? 0 : 1;
        if (mover == 0) s.bb0 |= mv; else s.bb1 |= mv;
        s.aux = mover;
        s.turn = -s.turn;
    }
    // -1 not terminal, 0 draw, 1 P1 wins, 2 P2 wins (check_winner then is_full, MCTS.h:279-288)
    __host__ __device__ static int result(const GameState &s)
    {
        if (s.aux >= 0 && four(s.aux == 0 ? s.bb0 : s.bb1)) return s.aux == 0 ? 1 : 2;
        if (__builtin_popcountll(s.bb0 | s.bb1) == CELLS) return 0;
        return -1;
    }
    __host__ __device__ static bool col_open(const GameState &s, int c)
    {
        return !(((s.bb0 | s.bb1) >> (c * BPC + ROWS - 1)) & 1ull);
    }
    __host__ __device__ static int num_valid(const GameState &s)    // Connect4.h:209-218
    {
        int n = 0;
#pragma unroll
        for (int c = 0; c < COLS; ++c) n += col_open(s, c) ? 1 : 0;
        return n;
    }
    __host__ __device__ static int nth_valid(const GameState &s, int n)
    {
        int k = 0, a = -1;
#pragma unroll
        for (int c = 0; c < COLS; ++c)
            if (col_open(s, c)) { if (k == n) a = c; ++k; }
        return a;
    }
    __host__ __device__ static float aux_utility(float child_m, float parent_m, float child_q, const SearchParams &p)
    {                                                      // Connect4.h:231-239
        if (!(p.mlh_slope > 0.0f)) return 0.0f;
        const float v = p.mlh_slope * (child_m - parent_m);
        const float lo = -p.mlh_cap, hi = p.mlh_cap;
        const float cl = (v < lo) ? lo : ((hi < v) ? hi : v);
        return cl * child_q;
    }
    __host__ __device__ static float terminal_aux(const GameState &, const SearchParams &) { return 0.0f; }
    __host__ __device__ static int sym_of_choice(int choice) { return choice; }
    // policy entry (evaluator frame) that belongs to `action` of the unsymmetrised leaf
    __host__ __device__ static int policy_index(int sym, int action) { return sym ? (COLS - 1 - action) : action; }
    // stone at display cell (row-major) of the symmetrised board (Connect4.h:249-280)
    __host__ __device__ static int cell_value(const GameState &s, int sym, int cell)
    {
        const int r = cell / COLS, c = cell - r * COLS;
        const int cs = sym ? (COLS - 1 - c) : c;
        const int bit = cs * BPC + (ROWS - 1 - r);
        return ((s.bb0 >> bit) & 1ull) ? 1 : (((s.bb1 >> bit) & 1ull) ? -1 : 0);
    }
    // is action index `a` of the symmetrised leaf legal
    __host__ __device__ static bool valid_in_frame(const GameState &s, int sym, int a)
    {
        return col_open(s, sym ? (COLS - 1 - a) : a);
    }
};

// ============================================================================ Othello
// src/cpp/Othello.h: bit i = row i/8, col i%8; actions 0-63 squares, 64 pass.
struct OthelloDev {
    static constexpr int GAME_ID = 1;
    static constexpr int LANES = 64, ACTIONS = 65, ROWS = 8, COLS = 8, CELLS = 64;
    static constexpr int MAX_PATH = OT_MAX_PATH, STATS = 6 + 8 * ACTIONS;
    static constexpr int SYM_CHOICES = 4;                 // MCTS_SYMMETRY_IDS {0,2,6,7} (Othello.h:45)
    static constexpr bool AUX_PLUS_ONE = false, AUX_NEGATE = true;   // Othello.h:31-32
    static constexpr int PASS = 64;
    static constexpr uint64_t NOT_A = 0xFEFEFEFEFEFEFEFEull, NOT_H = 0x7F7F7F7F7F7F7F7Full;

    template <int D>
    __host__ __device__ static uint64_t shift(uint64_t b)           // Othello.h:133-148
    {
        if (D == 0) return b >> 8;
        if (D == 1) return (b >> 7) & NOT_A;
        if (D == 2) return (b << 1) & NOT_A;
        if (D == 3) return (b << 9) & NOT_A;
        if (D == 4) return b << 8;
        if (D == 5) return (b << 7) & NOT_H;
        if (D == 6) return (b >> 1) & NOT_H;
        return (b >> 9) & NOT_H;
    }
    template <int D>
    __host__ __device__ static uint64_t valid_dir(uint64_t own, uint64_t opp, uint64_t empty)
    {
        uint64_t c = shift<D>(own) & opp;
#pragma unroll
        for (int i = 0; i < 5; ++i) c |= shift<D>(c) & opp;
        return shift<D>(c) & empty;
    }
    __host__ __device__ static uint64_t valid_positions(const GameState &s)   // Othello.h:155-171
    {
        const uint64_t own = (s.turn == 1) ? s.bb0 : s.bb1, opp = (s.turn == 1) ? s.bb1 : s.bb0;
        const uint64_t empty = ~(own | opp);
        return valid_dir<0>(own, opp, empty) | valid_dir<1>(own, opp, empty) | valid_dir<2>(own, opp, empty) |
               valid_dir<3>(own, opp, empty) | valid_dir<4>(own, opp, empty) | valid_dir<5>(own, opp, empty) |
               valid_dir<6>(own, opp, empty) | valid_dir<7>(own, opp, empty);
    }
    template <int D>
    __host__ __device__ static uint64_t flips_dir(uint64_t placed, uint64_t own, uint64_t opp)
    {
        uint64_t cand = 0, sq = shift<D>(placed);
#pragma unroll
        for (int i = 0; i < 6; ++i) {                      // a line holds at most 6 opponent stones
            if (!(sq & opp)) break;
            cand |= sq;
            sq = shift<D>(sq);
        }
        return (sq & own) ? cand : 0ull;
    }
    __host__ __device__ static int root_aux(uint64_t, uint64_t) { return 0; }   // import forgets passes (Othello.h:108-110)
    __host__ __device__ static void start(GameState &s)            // Othello.h:62-75
    {
        s.bb0 = (1ull << 28) | (1ull << 35); s.bb1 = (1ull << 27) | (1ull << 36); s.turn = 1; s.aux = 0;
    }
    __host__ __device__ static void import_cells(const int8_t *b, GameState &s)  // Othello.h:87-111
    {
        uint64_t bb0 = 0, bb1 = 0;
        for (int i = 0; i < CELLS; ++i) {
            const int8_t v = b[i];
            if (v == 1) bb0 |= 1ull << i; else if (v == -1) bb1 |= 1ull << i;
        }
        s.bb0 = bb0; s.bb1 = bb1; s.aux = 0;
    }
    __host__ __device__ static void step(GameState &s, int action)  // Othello.h:206-235
    {
        if (action == PASS) { s.aux += 1; s.turn = -s.turn; return; }
        const bool p1 = s.turn == 1;
        const uint64_t own = p1 ? s.bb0 : s.bb1, opp = p1 ? s.bb1 : s.bb0, placed = 1ull << action;
        const uint64_t f = flips_dir<0>(placed, own, opp) | flips_dir<1>(placed, own, opp) |
                           flips_dir<2>(placed, own, opp) | flips_dir<3>(placed, own, opp) |
                           flips_dir<4>(placed, own, opp) | flips_dir<5>(placed, own, opp) |
                           flips_dir<6>(placed, own, opp) | flips_dir<7>(placed, own, opp);
        const uint64_t nown = own | placed | f, nopp = opp & ~f;
        s.bb0 = p1 ? nown : nopp;
        s.bb1 = p1 ? nopp : nown;
        s.aux = 0;
        s.turn = -s.turn;
    }
    __host__ __device__ static bool over(const GameState &s) { return __builtin_popcountll(s.bb0 | s.bb1) == 64 || s.aux >= 2; }
    __host__ __device__ static int result(const GameState &s)       // Othello.h:241-258, is_full == is_game_over
    {
        if (!over(s)) return -1;
        const int a = __builtin_popcountll(s.bb0), b = __builtin_popcountll(s.bb1);
        return a > b ? 1 : (b > a ? 2 : 0);
    }
    __host__ __device__ static int num_valid(const GameState &s)    // Othello.h:283-294
    {
        if (over(s)) return 0;
        const uint64_t v = valid_positions(s);
        return v ? __builtin_popcountll(v) : 1;
    }
    __host__ __device__ static int nth_valid(const GameState &s, int n)
    {
        if (over(s)) return -1;
        uint64_t v = valid_positions(s);
        if (v == 0) return n == 0 ? PASS : -1;
        if (n >= static_cast<int>(__builtin_popcountll(v))) return -1;
        for (int i = 0; i < n; ++i) v &= v - 1;
        return __builtin_ffsll(static_cast<long long>(v)) - 1;
    }
    // MCTS.h:197-198 negates the child's mean before Othello.h:268-274 weighs it
    __host__ __device__ static float aux_utility(float child_m, float, float, const SearchParams &p)
    {
        if (!(p.score_utility_factor > 0.0f)) return 0.0f;
        return p.score_utility_factor * (-child_m);
    }
    // Othello.h:260-266 through the host-tabulated atan (index = diff*turn + 64)
    __host__ __device__ static float terminal_aux(const GameState &s, const SearchParams &p)
    {
        const int diff = __builtin_popcountll(s.bb0) - __builtin_popcountll(s.bb1);
        return p.term_aux_tab[diff * s.turn + 64];
    }
    __host__ __device__ static int sym_of_choice(int choice) { return choice == 0 ? 0 : (choice == 1 ? 2 : (choice == 2 ? 6 : 7)); }
    __host__ __device__ static int transform_sq(int sym, int sq)    // Othello.h:312-326
    {
        const int r = sq >> 3, c = sq & 7;
        int nr = r, nc = c;
        switch (sym) {
        case 1: nr = c;     nc = 7 - r; break;
        case 2: nr = 7 - r; nc = 7 - c; break;
        case 3: nr = 7 - c; nc = r;     break;
        case 4: nr = r;     nc = 7 - c; break;
        case 5: nr = 7 - r; nc = c;     break;
        case 6: nr = c;     nc = r;     break;
        case 7: nr = 7 - c; nc = 7 - r; break;
        default: break;
        }
        return nr * 8 + nc;
    }
    __host__ __device__ static int inverse_sym(int sym) { return sym == 1 ? 3 : (sym == 3 ? 1 : sym); }
    // inverse_symmetry_policy (Othello.h:373-387): unsym[T_inv(i)] = policy[i]  =>  the entry
    // for `action` is policy[T_sym(action)]; pass is not moved
    __host__ __device__ static int policy_index(int sym, int action)
    {
        return (action == PASS || sym == 0) ? action : transform_sq(sym, action);
    }
    // apply_symmetry moves stone i to T_sym(i): the stone shown at cell j was at T_inv(j)
    __host__ __device__ static int cell_value(const GameState &s, int sym, int cell)
    {
        const int src = sym == 0 ? cell : transform_sq(inverse_sym(sym), cell);
        return ((s.bb0 >> src) & 1ull) ? 1 : (((s.bb1 >> src) & 1ull) ? -1 : 0);
    }
    __host__ __device__ static bool valid_in_frame(const GameState &s, int sym, int a)
    {
        if (over(s)) return false;
        const uint64_t v = valid_positions(s);
        if (a == PASS) return v == 0;
        const int src = sym == 0 ? a : transform_sq(inverse_sym(sym), a);
        return (v >> src) & 1ull;
    }
};

}  // namespace az
