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

}  // namespace

PYBIND11_MODULE(env_cpp, m)
{
    m.doc() = "Game environments (host-side position objects; drop-in for the reference's env_cpp)";
    register_connect4(m);
    register_othello(m);
}
