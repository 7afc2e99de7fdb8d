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

}  // namespace

PYBIND11_MODULE(env_cpp, m)
{
    m.doc() = "Game environments (host-side position objects; drop-in for the reference's env_cpp)";
    register_connect4(m);
}
