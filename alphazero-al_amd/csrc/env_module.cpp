// env_module.cpp - CPython extension `env_cpp`: the caller-facing game objects of the
// reference (src/cpp/env_bindings.cpp:17-25, env_common.h:133-249, env_connect4.h:19-66).
// `env_cpp.connect4.Env` is a single host-side position that game.py / player.py / the GUIs
// hold one of per game; the batched, HBM-resident positions the search runs on live in
// libaz_mcts.so.  State is two bitboards + side to move (Connect4.h:15-29 bit layout); the
// int8 display grid the reference also carries is produced on demand.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdint>
#include <random>
#include <sstream>
#include <stdexcept>
#include <vector>

namespace py = pybind11;

namespace {

struct C4Env {
    static constexpr int R = 6, C = 7, W = 7;   // W bits per column (6 cells + sentinel)
    uint64_t bb[2] = {0, 0};
    int turn = 1;
    int last = -1;

    void reset() { bb[0] = bb[1] = 0; turn = 1; last = -1; }
    int pieces() const { return __builtin_popcountll(bb[0] | bb[1]); }
    int height(int c) const { return __builtin_popcountll((bb[0] | bb[1]) & (0x7Full << (W * c))); }

    void step(int col)                                    // Connect4.h:159-172
    {
        const int p = turn == 1 ? 0 : 1;
        bb[p] |= 1ull << (W * col + height(col));
        last = p;
        turn = -turn;
    }
    int winner() const                                    // Connect4.h:182-203
    {
        if (last < 0) return 0;
        const uint64_t b = bb[last];
        for (int d : {1, 7, 6, 8}) {
            const uint64_t t = b & (b >> d);
            if (t & (t >> (2 * d))) return last == 0 ? 1 : -1;
        }
        return 0;
    }
    bool full() const { return pieces() == R * C; }
    bool open(int c) const { return height(c) < R; }
    int cell(int r, int c) const
    {
        const int bit = W * c + (R - 1 - r);
        return ((bb[0] >> bit) & 1) ? 1 : (((bb[1] >> bit) & 1) ? -1 : 0);
    }
    void import_grid(const int8_t *g)                     // Connect4.h:87-129
    {
        bb[0] = bb[1] = 0;
        for (int c = 0; c < C; ++c) {
            int h = 0;
            for (int r = R - 1; r >= 0; --r) {
                const int8_t v = g[r * C + c];
                if (v == 0) break;
                bb[v == 1 ? 0 : 1] |= 1ull << (W * c + h++);
            }
        }
        const int n = pieces();
        last = n == 0 ? -1 : ((n & 1) ? 0 : 1);
    }
    void mirror(int sym)                                  // Connect4.h:249-280
    {
        if (sym == 0) return;
        for (auto &b : bb) {
            uint64_t d = 0;
            for (int c = 0; c < C; ++c) d |= ((b >> (W * c)) & 0x7Full) << (W * (C - 1 - c));
            b = d;
        }
    }
};

py::array_t<float> board_of(const C4Env &e)               // env_common.h:36-50 (float32!)
{
    py::array_t<float> a({C4Env::R, C4Env::C});
    auto v = a.mutable_unchecked<2>();
    for (int r = 0; r < C4Env::R; ++r)
        for (int c = 0; c < C4Env::C; ++c) v(r, c) = static_cast<float>(e.cell(r, c));
    return a;
}

void set_board(C4Env &e, py::array_t<float, py::array::c_style | py::array::forcecast> arr)
{                                                         // env_common.h:55-70
    if (arr.ndim() != 2 || arr.shape(0) != C4Env::R || arr.shape(1) != C4Env::C)
        throw std::runtime_error("board shape must be (6, 7)");
    auto v = arr.unchecked<2>();
    int8_t g[C4Env::R * C4Env::C];
    for (int r = 0; r < C4Env::R; ++r)
        for (int c = 0; c < C4Env::C; ++c) g[r * C4Env::C + c] = static_cast<int8_t>(v(r, c));
    e.import_grid(g);
    e.turn = (e.pieces() % 2 == 0) ? 1 : -1;
}

py::array_t<float> current_state(const C4Env &e)          // env_common.h:93-119
{
    py::array_t<float> s({1, 3, C4Env::R, C4Env::C});
    auto v = s.mutable_unchecked<4>();
    for (int r = 0; r < C4Env::R; ++r)
        for (int c = 0; c < C4Env::C; ++c) {
            const int x = e.cell(r, c);
            v(0, 0, r, c) = (x == e.turn) ? 1.0f : 0.0f;
            v(0, 1, r, c) = (x == -e.turn) ? 1.0f : 0.0f;
            v(0, 2, r, c) = static_cast<float>(e.turn);
        }
    return s;
}

void register_connect4(py::module_ &m)
{
    auto sub = m.def_submodule("connect4", "Connect4 environment");
    py::class_<C4Env>(sub, "Env")
        .def(py::init<>())
        .def(py::init([](py::array_t<float, py::array::c_style | py::array::forcecast> b) {
                 C4Env e;
                 set_board(e, b);
                 return e;
             }),
             py::arg("board"))
        .def("reset", &C4Env::reset)
        .def("copy", [](const C4Env &e) { return C4Env(e); })
        .def("step", &C4Env::step, py::arg("action"))
        .def("winPlayer", &C4Env::winner)
        .def("check_winner", &C4Env::winner)
        .def("check_full", &C4Env::full)
        .def_property(
            "turn", [](const C4Env &e) { return e.turn; }, [](C4Env &e, int t) { e.turn = t; })
        .def_property_readonly_static("NUM_SYMMETRIES", [](py::object) { return 2; })
        .def(
            "apply_symmetry",
            [](C4Env &e, int sym_id, bool inplace) {
                if (inplace) { e.mirror(sym_id); return e; }
                C4Env c(e);
                c.mirror(sym_id);
                return c;
            },
            py::arg("sym_id"), py::arg("inplace") = false)
        .def("random_symmetry",
             [](const C4Env &e) {
                 static thread_local std::mt19937 rng(std::random_device{}());
                 const int sym = std::uniform_int_distribution<int>(0, 1)(rng);
                 C4Env c(e);
                 c.mirror(sym);
                 return py::make_tuple(c, sym);
             })
        .def_property("board", &board_of, &set_board)
        .def("valid_move",
             [](const C4Env &e) {
                 py::list l;
                 for (int c = 0; c < C4Env::C; ++c) if (e.open(c)) l.append(c);
                 return l;
             })
        .def("valid_mask",
             [](const C4Env &e) {
                 py::list l;
                 for (int c = 0; c < C4Env::C; ++c) l.append(e.open(c));
                 return l;
             })
        .def("current_state", &current_state)
        .def(py::pickle(
            [](const C4Env &e) { return py::make_tuple(board_of(e), e.turn); },
            [](py::tuple t) {
                if (t.size() != 2) throw std::runtime_error("Invalid pickle state");
                C4Env e;
                set_board(e, t[0].cast<py::array_t<float>>());
                e.turn = t[1].cast<int>();
                return e;
            }))
        .def("done", [](const C4Env &e) { return e.winner() != 0 || e.full(); })
        .def_static(
            "inverse_symmetry_action", [](int sym_id, int col) { return sym_id == 0 ? col : (C4Env::C - 1 - col); },
            py::arg("sym_id"), py::arg("col"))
        .def("show",
             [](const C4Env &e) {
                 std::ostringstream os;
                 os << "====================\n";
                 for (int r = 0; r < C4Env::R; ++r) {
                     for (int c = 0; c < C4Env::C; ++c) {
                         if (c) os << ' ';
                         const int v = e.cell(r, c);
                         os << (v == 0 ? '_' : (v == 1 ? 'X' : 'O'));
                     }
                     os << '\n';
                 }
                 os << "0 1 2 3 4 5 6\n====================";
                 py::print(os.str());
             })
        // extension: the bitboards, so that a batch of Env objects can be uploaded to HBM
        .def_property_readonly("bitboards", [](const C4Env &e) { return py::make_tuple(e.bb[0], e.bb[1]); });
}

// ============================================================================ Othello
// src/cpp/Othello.h + env_othello.h: bit i = row i/8, col i%8; action 64 = pass.
struct OtEnv {
    static constexpr int R = 8, C = 8, PASS = 64;
    static constexpr uint64_t NOT_A = 0xFEFEFEFEFEFEFEFEull, NOT_H = 0x7F7F7F7F7F7F7F7Full;
    uint64_t bb[2];
    int turn, passes, last;
    OtEnv() { reset(); }

    void reset()                                           // Othello.h:63-78
    {
        bb[0] = (1ull << 28) | (1ull << 35);
        bb[1] = (1ull << 27) | (1ull << 36);
        turn = 1; passes = 0; last = -1;
    }
    static uint64_t shift(uint64_t b, int d)               // Othello.h:133-148
    {
        switch (d) {
        case 0: return b >> 8;
        case 1: return (b >> 7) & NOT_A;
        case 2: return (b << 1) & NOT_A;
        case 3: return (b << 9) & NOT_A;
        case 4: return b << 8;
        case 5: return (b << 7) & NOT_H;
        case 6: return (b >> 1) & NOT_H;
        default: return (b >> 9) & NOT_H;
        }
    }
    int pieces() const { return __builtin_popcountll(bb[0] | bb[1]); }
    uint64_t valid_positions() const                       // Othello.h:155-171
    {
        const int p = turn == 1 ? 0 : 1;
        const uint64_t own = bb[p], opp = bb[1 - p], empty = ~(own | opp);
        uint64_t v = 0;
        for (int d = 0; d < 8; ++d) {
            uint64_t c = shift(own, d) & opp;
            for (int i = 0; i < 5; ++i) c |= shift(c, d) & opp;
            v |= shift(c, d) & empty;
        }
        return v;
    }
    void step(int action)                                  // Othello.h:206-235
    {
        if (action == PASS) { ++passes; turn = -turn; return; }
        const int p = turn == 1 ? 0 : 1;
        const uint64_t own = bb[p], opp = bb[1 - p], placed = 1ull << action;
        uint64_t flips = 0;
        for (int d = 0; d < 8; ++d) {
            uint64_t cand = 0, sq = shift(placed, d);
            while (sq & opp) { cand |= sq; sq = shift(sq, d); }
            if (sq & own) flips |= cand;
        }
        bb[p] |= placed | flips;
        bb[1 - p] &= ~flips;
        passes = 0; last = p; turn = -turn;
    }
    bool over() const { return pieces() == 64 || passes >= 2; }   // Othello.h:241-244
    int winner() const                                     // Othello.h:250-258
    {
        if (!over()) return 0;
        const int a = __builtin_popcountll(bb[0]), b = __builtin_popcountll(bb[1]);
        return a > b ? 1 : (b > a ? -1 : 0);
    }
    std::vector<int> moves() const                         // Othello.h:283-294
    {
        std::vector<int> m;
        if (over()) return m;
        uint64_t v = valid_positions();
        if (!v) { m.push_back(PASS); return m; }
        for (; v; v &= v - 1) m.push_back(__builtin_ctzll(v));
        return m;
    }
    int cell(int r, int c) const
    {
        const int bit = r * 8 + c;
        return ((bb[0] >> bit) & 1) ? 1 : (((bb[1] >> bit) & 1) ? -1 : 0);
    }
    static void transform(int sym, int r, int c, int &nr, int &nc)   // Othello.h:312-326
    {
        switch (sym) {
        case 1: nr = c;     nc = 7 - r; break;
        case 2: nr = 7 - r; nc = 7 - c; break;
        case 3: nr = 7 - c; nc = r;     break;
        case 4: nr = r;     nc = 7 - c; break;
        case 5: nr = 7 - r; nc = c;     break;
        case 6: nr = c;     nc = r;     break;
        case 7: nr = 7 - c; nc = 7 - r; break;
        default: nr = r;    nc = c;     break;
        }
    }
    static int inverse_sym(int s) { return s == 1 ? 3 : (s == 3 ? 1 : s); }
    void apply_sym(int sym)                                // Othello.h:329-353
    {
        if (sym == 0) return;
        for (auto &b : bb) {
            uint64_t r = 0;
            for (uint64_t bits = b; bits; bits &= bits - 1) {
                const int i = __builtin_ctzll(bits);
                int nr, nc;
                transform(sym, i / 8, i % 8, nr, nc);
                r |= 1ull << (nr * 8 + nc);
            }
            b = r;
        }
    }
    void import_grid(const int8_t *g)                      // Othello.h:87-111
    {
        bb[0] = bb[1] = 0;
        for (int i = 0; i < 64; ++i) {
            if (g[i] == 1) bb[0] |= 1ull << i;
            else if (g[i] == -1) bb[1] |= 1ull << i;
        }
        passes = 0; last = -1;
    }
};

template <class E>
py::array_t<float> grid_of(const E &e)                    // env_common.h:36-50
{
    py::array_t<float> a({E::R, E::C});
    auto v = a.template mutable_unchecked<2>();
    for (int r = 0; r < E::R; ++r)
        for (int c = 0; c < E::C; ++c) v(r, c) = static_cast<float>(e.cell(r, c));
    return a;
}

void ot_set_board(OtEnv &e, py::array_t<float, py::array::c_style | py::array::forcecast> arr)
{                                                         // env_common.h:55-70
    if (arr.ndim() != 2 || arr.shape(0) != 8 || arr.shape(1) != 8) throw std::runtime_error("board shape must be (8, 8)");
    auto v = arr.unchecked<2>();
    int8_t g[64];
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) g[r * 8 + c] = static_cast<int8_t>(v(r, c));
    e.import_grid(g);
    e.turn = (e.pieces() % 2 == 0) ? 1 : -1;
}

py::array_t<float> ot_current_state(const OtEnv &e)       // env_common.h:93-119
{
    py::array_t<float> s({1, 3, 8, 8});
    auto v = s.mutable_unchecked<4>();
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) {
            const int x = e.cell(r, c);
            v(0, 0, r, c) = (x == e.turn) ? 1.0f : 0.0f;
            v(0, 1, r, c) = (x == -e.turn) ? 1.0f : 0.0f;
            v(0, 2, r, c) = static_cast<float>(e.turn);
        }
    return s;
}

void register_othello(py::module_ &m)
{
    auto sub = m.def_submodule("othello", "Othello environment");
    py::class_<OtEnv>(sub, "Env")
        .def(py::init<>())
        .def(py::init([](py::array_t<float, py::array::c_style | py::array::forcecast> b) {
                 OtEnv e;
                 ot_set_board(e, b);
                 return e;
             }),
             py::arg("board"))
        .def("reset", &OtEnv::reset)
        .def("copy", [](const OtEnv &e) { return OtEnv(e); })
        .def("step", &OtEnv::step, py::arg("action"))
        .def("winPlayer", &OtEnv::winner)
        .def("check_winner", &OtEnv::winner)
        .def("check_full", &OtEnv::over)
        .def_property(
            "turn", [](const OtEnv &e) { return e.turn; }, [](OtEnv &e, int t) { e.turn = t; })
        .def_property_readonly_static("NUM_SYMMETRIES", [](py::object) { return 8; })
        .def(
            "apply_symmetry",
            [](OtEnv &e, int sym_id, bool inplace) {
                if (inplace) { e.apply_sym(sym_id); return e; }
                OtEnv c(e);
                c.apply_sym(sym_id);
                return c;
            },
            py::arg("sym_id"), py::arg("inplace") = false)
        .def("random_symmetry",
             [](const OtEnv &e) {
                 static thread_local std::mt19937 rng(std::random_device{}());
                 const int sym = std::uniform_int_distribution<int>(0, 7)(rng);
                 OtEnv c(e);
                 c.apply_sym(sym);
                 return py::make_tuple(c, sym);
             })
        .def_property("board", &grid_of<OtEnv>, &ot_set_board)
        .def("valid_move",
             [](const OtEnv &e) {
                 py::list l;
                 for (int a : e.moves()) l.append(a);
                 return l;
             })
        .def("valid_mask",
             [](const OtEnv &e) {
                 bool mask[65] = {};
                 for (int a : e.moves()) mask[a] = true;
                 py::list l;
                 for (bool b : mask) l.append(b);
                 return l;
             })
        .def("current_state", &ot_current_state)
        .def(py::pickle(
            [](const OtEnv &e) { return py::make_tuple(grid_of(e), e.turn); },
            [](py::tuple t) {
                if (t.size() != 2) throw std::runtime_error("Invalid pickle state");
                OtEnv e;
                ot_set_board(e, t[0].cast<py::array_t<float>>());
                e.turn = t[1].cast<int>();
                return e;
            }))
        .def("done", &OtEnv::over)
        .def_static(
            "inverse_symmetry_action",
            [](int sym_id, int action) {                  // env_othello.h:42-52
                if (action == OtEnv::PASS || sym_id == 0) return action;
                int nr, nc;
                OtEnv::transform(OtEnv::inverse_sym(sym_id), action / 8, action % 8, nr, nc);
                return nr * 8 + nc;
            },
            py::arg("sym_id"), py::arg("action"))
        .def("show",
             [](const OtEnv &e) {
                 std::ostringstream os;
                 os << "========================\n  0 1 2 3 4 5 6 7\n";
                 for (int r = 0; r < 8; ++r) {
                     os << r << ' ';
                     for (int c = 0; c < 8; ++c) {
                         if (c) os << ' ';
                         const int v = e.cell(r, c);
                         os << (v == 0 ? '.' : (v == 1 ? 'X' : 'O'));
                     }
                     os << '\n';
                 }
                 os << "========================";
                 py::print(os.str());
             })
        .def_property_readonly("bitboards", [](const OtEnv &e) { return py::make_tuple(e.bb[0], e.bb[1]); });
}


// ============================================================================ Gomoku
// Host-side position object only: the reference binds no Gomoku search (mcts_bindings.cpp:393-394
// registers Connect4 and Othello), so this exists for the surface `src.env_cpp.gomoku.Env`
// (env_gomoku.h:67-170 over Gomoku.h).  Square board of run-time size, n stones in a row win.
struct GmEnv {
    int n = 15, need = 5;
    std::vector<int8_t> cells;
    int turn = 1, stones = 0, last = -1, winner = 0;
    bool over = false;

    GmEnv(int board_size = 15, int n_in_row = 5) { configure(board_size, n_in_row); }

    void configure(int board_size, int n_in_row)
    {
        // messages as Gomoku.h:214-222
        if (board_size <= 0) throw std::runtime_error("board_size must be positive");
        if (n_in_row <= 1) throw std::runtime_error("n_in_row must be >= 2");
        if (n_in_row > board_size) throw std::runtime_error("n_in_row must be <= board size");
        n = board_size;
        need = n_in_row;
        cells.assign(static_cast<size_t>(n) * n, 0);
        clear();
    }
    void clear()
    {
        std::fill(cells.begin(), cells.end(), 0);
        turn = 1; stones = 0; last = -1; winner = 0; over = false;
    }
    int area() const { return n * n; }
    bool inside(int r, int c) const { return r >= 0 && r < n && c >= 0 && c < n; }
    // length of the run of `who` through (r, c) along (dr, dc), the cell itself included
    int run(int r, int c, int dr, int dc, int who) const
    {
        int len = 1;
        for (int s = -1; s <= 1; s += 2)
            for (int k = 1; inside(r + s * k * dr, c + s * k * dc) && cells[(r + s * k * dr) * n + c + s * k * dc] == who; ++k) ++len;
        return len;
    }
    bool wins_through(int a, int who) const
    {
        const int r = a / n, c = a % n;
        return run(r, c, 1, 0, who) >= need || run(r, c, 0, 1, who) >= need || run(r, c, 1, 1, who) >= need ||
               run(r, c, 1, -1, who) >= need;
    }
    void play(int a)                                   // Gomoku.h:73-99
    {
        if (over) throw std::runtime_error("game is already finished");
        if (a < 0 || a >= area()) throw std::runtime_error("action out of range");
        if (cells[a] != 0) throw std::runtime_error("cell is already occupied");
        cells[a] = static_cast<int8_t>(turn);
        ++stones;
        last = a;
        if (wins_through(a, turn)) { winner = turn; over = true; }
        else if (stones == area()) { winner = 0; over = true; }
        turn = -turn;
    }
    void set_turn(int t)
    {
        if (t != 1 && t != -1) throw std::runtime_error("turn must be 1 or -1");
        turn = t;
    }
    // rebuild everything from the grid: side to move from the stone counts, first winning
    // stone in scan order decides the winner (Gomoku.h:166-209)
    void rescan()
    {
        int p1 = 0, p2 = 0;
        last = -1;
        for (int i = 0; i < area(); ++i) {
            if (cells[i] == 1) { ++p1; last = i; }
            else if (cells[i] == -1) { ++p2; last = i; }
            else if (cells[i] != 0) throw std::runtime_error("board values must be -1, 0, or 1");
        }
        stones = p1 + p2;
        turn = p1 == p2 ? 1 : (p1 == p2 + 1 ? -1 : (stones % 2 == 0 ? 1 : -1));
        winner = 0;
        for (int i = 0; i < area() && winner == 0; ++i)
            if (cells[i] != 0 && wins_through(i, cells[i])) winner = cells[i];
        over = winner != 0 || stones == area();
    }
    // the dihedral group of the square, numbered as Gomoku.h:270-289
    void map(int sym, int r, int c, int &nr, int &nc) const
    {
        const int m = n - 1;
        switch (sym) {
        case 0: nr = r; nc = c; break;
        case 1: nr = c; nc = m - r; break;
        case 2: nr = m - r; nc = m - c; break;
        case 3: nr = m - c; nc = r; break;
        case 4: nr = r; nc = m - c; break;
        case 5: nr = m - r; nc = c; break;
        case 6: nr = c; nc = r; break;
        case 7: nr = m - c; nc = m - r; break;
        default: throw std::runtime_error("invalid symmetry id");
        }
    }
    int map_action(int sym, int a) const
    {
        int nr, nc;
        map(sym, a / n, a % n, nr, nc);
        return nr * n + nc;
    }
    void transform(int sym)
    {
        if (sym < 0 || sym >= 8) throw std::runtime_error("invalid symmetry id");
        if (sym == 0) return;
        std::vector<int8_t> out(cells.size(), 0);
        for (int a = 0; a < area(); ++a) out[map_action(sym, a)] = cells[a];
        cells.swap(out);
        if (last >= 0) last = map_action(sym, last);
    }
};

py::array_t<float> gm_grid(const GmEnv &e)
{
    py::array_t<float> arr({e.n, e.n});
    auto b = arr.mutable_unchecked<2>();
    for (int r = 0; r < e.n; ++r)
        for (int c = 0; c < e.n; ++c) b(r, c) = static_cast<float>(e.cells[r * e.n + c]);
    return arr;
}

void gm_set_grid(GmEnv &e, py::array_t<float, py::array::c_style | py::array::forcecast> arr)
{
    auto b = arr.unchecked<2>();
    if (b.shape(0) != e.n || b.shape(1) != e.n) throw std::runtime_error("board shape does not match environment dimensions");
    for (int r = 0; r < e.n; ++r)
        for (int c = 0; c < e.n; ++c) e.cells[r * e.n + c] = static_cast<int8_t>(b(r, c));
    e.rescan();
}

GmEnv gm_from_grid(py::array_t<float, py::array::c_style | py::array::forcecast> arr, int n_in_row)
{
    auto b = arr.unchecked<2>();
    if (b.shape(0) != b.shape(1)) throw std::runtime_error("board must be square");
    GmEnv e(static_cast<int>(b.shape(0)), n_in_row);
    gm_set_grid(e, arr);
    return e;
}

void register_gomoku(py::module_ &m)
{
    auto sub = m.def_submodule("gomoku", "Gomoku-like environment with configurable board and win length");
    py::class_<GmEnv>(sub, "Env")
        .def(py::init<int, int>(), py::arg("board_size") = 15, py::arg("n_in_row") = 5)
        .def(py::init(&gm_from_grid), py::arg("board"), py::arg("n_in_row") = 5)
        .def("reset", &GmEnv::clear)
        .def("copy", [](const GmEnv &e) { return GmEnv(e); })
        .def("step", &GmEnv::play, py::arg("action"))
        .def("winPlayer", [](const GmEnv &e) { return e.winner; })
        .def("check_winner", [](const GmEnv &e) { return e.winner; })
        .def("check_full", [](const GmEnv &e) { return e.stones == e.area(); })
        .def_property("turn", [](const GmEnv &e) { return e.turn; }, &GmEnv::set_turn)
        .def_property_readonly_static("NUM_SYMMETRIES", [](py::object) { return 8; })
        .def(
            "apply_symmetry",
            [](GmEnv &e, int sym_id, bool inplace) {
                if (inplace) { e.transform(sym_id); return e; }
                GmEnv c(e);
                c.transform(sym_id);
                return c;
            },
            py::arg("sym_id"), py::arg("inplace") = false)
        .def("random_symmetry",
             [](const GmEnv &e) {
                 static thread_local std::mt19937 rng(std::random_device{}());
                 const int sym = std::uniform_int_distribution<int>(0, 7)(rng);
                 GmEnv c(e);
                 c.transform(sym);
                 return py::make_tuple(c, sym);
             })
        .def_property("board", &gm_grid, &gm_set_grid)
        .def_property_readonly("board_size", [](const GmEnv &e) { return e.n; })
        .def_property_readonly("rows", [](const GmEnv &e) { return e.n; })
        .def_property_readonly("cols", [](const GmEnv &e) { return e.n; })
        .def_property_readonly("n_in_row", [](const GmEnv &e) { return e.need; })
        .def_property_readonly("action_size", &GmEnv::area)
        .def_property_readonly("num_symmetries", [](const GmEnv &) { return 8; })
        .def("set_params", &GmEnv::configure, py::arg("board_size"), py::arg("n_in_row"))
        .def("done", [](const GmEnv &e) { return e.over; })
        .def(
            "coord_to_action",
            [](const GmEnv &e, int row, int col) {
                if (!e.inside(row, col)) throw std::runtime_error("row/col out of range");
                return row * e.n + col;
            },
            py::arg("row"), py::arg("col"))
        .def(
            "step_xy",
            [](GmEnv &e, int row, int col) {
                if (!e.inside(row, col)) throw std::runtime_error("row/col out of range");
                e.play(row * e.n + col);
            },
            py::arg("row"), py::arg("col"))
        .def(
            "action_to_coord",
            [](const GmEnv &e, int action) {
                if (action < 0 || action >= e.area()) throw std::runtime_error("action out of range");
                return py::make_tuple(action / e.n, action % e.n);
            },
            py::arg("action"))
        .def("valid_move",
             [](const GmEnv &e) {
                 py::list l;
                 for (int a = 0; a < e.area(); ++a)
                     if (e.cells[a] == 0) l.append(a);
                 return l;
             })
        .def("valid_mask",
             [](const GmEnv &e) {
                 py::list l;
                 for (int a = 0; a < e.area(); ++a) l.append(e.cells[a] == 0);
                 return l;
             })
        .def("current_state",
             [](const GmEnv &e) {                         // env_common.h:93-119
                 py::array_t<float> st({1, 3, e.n, e.n});
                 auto b = st.mutable_unchecked<4>();
                 for (int r = 0; r < e.n; ++r)
                     for (int c = 0; c < e.n; ++c) {
                         const int v = e.cells[r * e.n + c];
                         b(0, 0, r, c) = v == e.turn ? 1.0f : 0.0f;
                         b(0, 1, r, c) = v == -e.turn ? 1.0f : 0.0f;
                         b(0, 2, r, c) = static_cast<float>(e.turn);
                     }
                 return st;
             })
        .def(
            "inverse_symmetry_action",
            [](const GmEnv &e, int sym_id, int action) {  // (sic) the forward map, as Gomoku.h:128-141
                if (action < 0 || action >= e.area()) throw std::runtime_error("action out of range");
                if (sym_id < 0 || sym_id >= 8) throw std::runtime_error("invalid symmetry id");
                return e.map_action(sym_id, action);
            },
            py::arg("sym_id"), py::arg("action"))
        .def("show",
             [](const GmEnv &e) {
                 std::ostringstream os;
                 os << "==============================\n    ";
                 for (int c = 0; c < e.n; ++c) os << c % 10 << ' ';
                 os << '\n';
                 for (int r = 0; r < e.n; ++r) {
                     os << (r < 10 ? " " : "") << r << "  ";
                     for (int c = 0; c < e.n; ++c) {
                         const int v = e.cells[r * e.n + c];
                         os << (v == 0 ? '.' : (v == 1 ? 'X' : 'O')) << ' ';
                     }
                     os << '\n';
                 }
                 os << "==============================";
                 py::print(os.str());
             })
        .def(py::pickle(
            [](const GmEnv &e) { return py::make_tuple(gm_grid(e), e.turn, e.need); },
            [](py::tuple t) {
                if (t.size() != 3) throw std::runtime_error("Invalid pickle state");
                GmEnv e = gm_from_grid(t[0].cast<py::array_t<float>>(), t[2].cast<int>());
                e.set_turn(t[1].cast<int>());
                return e;
            }));
}

}  // namespace

PYBIND11_MODULE(env_cpp, m)
{
    m.doc() = "Game environments (host-side position objects; drop-in for the reference's env_cpp)";
    register_connect4(m);
    register_othello(m);
    register_gomoku(m);
}
