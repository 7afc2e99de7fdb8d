// mcts_module.cpp - CPython extension `mcts_cpp`: the reference's Python-visible search
// surface (src/cpp/mcts_bindings.cpp:31-395) re-exposed on top of the C ABI of
// include/az_mcts.h.  Same class / method / keyword names, same dtypes, shapes and error
// behaviour, so `from src import mcts_cpp` in the reference's src/MCTS_cpp.py, player.py and
// GUI code binds to the HIP engine unchanged.  This file holds no search logic.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "az_mcts.h"

namespace py = pybind11;

namespace {

void check(int rc)
{
    if (rc != AZ_OK) throw std::runtime_error(az_last_error());   // -> Python RuntimeError
}

template <class T>
using carray = py::array_t<T, py::array::c_style | py::array::forcecast>;

// One engine per Python object; `game` selects the C-ABI game id.
template <int GAME>
class Batched {
public:
    explicit Batched(int n_envs) { check(az_mcts_create(GAME, n_envs, -1, &h_)); }
    ~Batched() { az_mcts_destroy(h_); }
    Batched(const Batched &) = delete;
    Batched &operator=(const Batched &) = delete;

    az_mcts *h() const { return h_; }
    int n() const { return az_mcts_num_envs(h_); }
    static int action_size() { return az_game_action_size(GAME); }
    static int board_size() { return az_game_board_size(GAME); }
    static std::vector<py::ssize_t> board_shape()
    {
        return {az_game_board_rows(GAME), az_game_board_cols(GAME)};
    }

private:
    az_mcts *h_ = nullptr;
};

std::vector<py::ssize_t> leaf_shape(py::ssize_t n, const std::vector<py::ssize_t> &board)
{
    std::vector<py::ssize_t> s{n};
    s.insert(s.end(), board.begin(), board.end());
    return s;
}

template <int GAME>
void register_game(py::module_ &m, const char *suffix)
{
    using BM = Batched<GAME>;
    const std::string cls = std::string("BatchedMCTS_") + suffix;
    // IEvaluator_<G> / RolloutEvaluator_<G> (mcts_bindings.cpp:41-48): the rollout evaluator is
    // a tag object here - the random playouts run inside the engine.
    // `device_rng` (an addition): False = playout moves from the reference's random stream on the host
    // (bit-exact, az_mcts_search_rollout), True = playouts on the device (az_mcts_search_rollout_dev).
    struct IEval { bool device_rng = false; };
    struct RolloutEval : IEval {};
    py::class_<IEval>(m, (std::string("IEvaluator_") + suffix).c_str());
    py::class_<RolloutEval, IEval>(m, (std::string("RolloutEvaluator_") + suffix).c_str())
        .def(py::init<>())
        .def_readwrite("device_rng", &RolloutEval::device_rng);

    py::class_<BM>(m, cls.c_str())
        .def(py::init<int>(), py::arg("n_envs"))
        // live reference to the engine's config, setter copies (mcts_bindings.cpp:55-58)
        .def_property(
            "config",
            [](BM &self) -> az_search_config & { return *az_mcts_config(self.h()); },
            [](BM &self, const az_search_config &c) { *az_mcts_config(self.h()) = c; },
            py::return_value_policy::reference_internal)
        .def("set_seed", [](BM &self, int seed) { check(az_mcts_set_seed(self.h(), seed)); },
             "Set random seed")
        .def("reset_env", [](BM &self, int env) { check(az_mcts_reset_env(self.h(), env)); },
             "Reset MCTS tree for specified environment index")
        .def("get_num_envs", &BM::n, "Return number of parallel environments")
        .def("prune_roots",
             [](BM &self, carray<int> actions) {
                 py::buffer_info b = actions.request();
                 if (b.ndim != 1) throw std::runtime_error("Actions must be 1D array");
                 const int *p = static_cast<const int *>(b.ptr);
                 int rc;
                 {
                     py::gil_scoped_release rel;
                     rc = az_mcts_prune_roots(self.h(), p, b.size);
                 }
                 check(rc);
             })
        .def("search_batch",
             [](BM &self, carray<int8_t> input_boards, carray<int> turns) {
                 auto bi = input_boards.request();
                 auto bt = turns.request();
                 const py::ssize_t n = bi.ndim ? bi.shape[0] : 0;
                 if (n != self.n())
                     throw std::runtime_error("search_batch: input_boards batch size (" + std::to_string(n) +
                                              ") must match n_envs (" + std::to_string(self.n()) + ")");
                 if (bt.size != n) throw std::runtime_error("Turns size must match batch size");
                 if (bi.size != n * BM::board_size()) throw std::runtime_error("search_batch: bad board shape");
                 py::array_t<int8_t> ob(leaf_shape(n, BM::board_shape()));
                 py::array_t<float> d(n), p1(n), p2(n);
                 py::array_t<uint8_t> it(n);
                 py::array_t<int> ot(n);
                 py::array_t<uint8_t> vm(std::vector<py::ssize_t>{n, BM::action_size()});
                 int rc;
                 {
                     auto *pin = static_cast<const int8_t *>(bi.ptr);
                     auto *ptn = static_cast<const int32_t *>(bt.ptr);
                     auto *pob = ob.mutable_data(); auto *pd = d.mutable_data();
                     auto *pp1 = p1.mutable_data(); auto *pp2 = p2.mutable_data();
                     auto *pit = it.mutable_data(); auto *pot = ot.mutable_data();
                     auto *pvm = vm.mutable_data();
                     py::gil_scoped_release rel;
                     rc = az_mcts_search_batch(self.h(), pin, ptn, n, pob, pd, pp1, pp2, pit, pot, pvm);
                 }
                 check(rc);
                 return py::make_tuple(ob, d, p1, p2, it, ot, vm);
             })
        .def("backprop_batch",
             [](BM &self, carray<float> policy_logits, carray<float> d_vals, carray<float> p1w_vals,
                carray<float> p2w_vals, carray<float> moves_left, carray<uint8_t> is_term) {
                 auto bp = policy_logits.request();
                 const py::ssize_t n = self.n();
                 const py::ssize_t got = bp.ndim ? bp.shape[0] : 0;
                 if (got != n)
                     throw std::runtime_error("backprop_batch: policy_logits batch size (" + std::to_string(got) +
                                              ") must match n_envs (" + std::to_string(n) + ")");
                 if (bp.size != n * BM::action_size()) throw std::runtime_error("backprop_batch: bad policy shape");
                 if (d_vals.size() != n || p1w_vals.size() != n || p2w_vals.size() != n)
                     throw std::runtime_error("backprop_batch: d/p1w/p2w size must match n_envs (" + std::to_string(n) + ")");
                 if (moves_left.size() != n)
                     throw std::runtime_error("backprop_batch: moves_left size (" + std::to_string(moves_left.size()) +
                                              ") must match n_envs (" + std::to_string(n) + ")");
                 if (is_term.size() != n)
                     throw std::runtime_error("backprop_batch: is_term size (" + std::to_string(is_term.size()) +
                                              ") must match n_envs (" + std::to_string(n) + ")");
                 int rc;
                 {
                     auto *a = policy_logits.data(); auto *b = d_vals.data(); auto *c = p1w_vals.data();
                     auto *e = p2w_vals.data(); auto *f = moves_left.data(); auto *g = is_term.data();
                     py::gil_scoped_release rel;
                     rc = az_mcts_backprop_batch(self.h(), a, b, c, e, f, g, n);
                 }
                 check(rc);
             },
             py::arg("policy_logits"), py::arg("d_vals"), py::arg("p1w_vals"), py::arg("p2w_vals"),
             py::arg("moves_left"), py::arg("is_term"))
        .def("remove_all_vl",
             [](BM &self, int K) {
                 int rc;
                 {
                     py::gil_scoped_release rel;
                     rc = az_mcts_remove_all_vl(self.h(), K);
                 }
                 check(rc);
             },
             py::arg("K"), "Remove all VL (n_inflight) from all trees. For exception-safety cleanup.")
        .def("search_batch_vl",
             [](BM &self, int K, carray<int8_t> input_boards, carray<int> turns) {
                 auto bi = input_boards.request();
                 auto bt = turns.request();
                 const py::ssize_t n = self.n();
                 const py::ssize_t got = bi.ndim ? bi.shape[0] : 0;
                 if (got != n)
                     throw std::runtime_error("search_batch_vl: input batch (" + std::to_string(got) +
                                              ") != n_envs (" + std::to_string(n) + ")");
                 if (bt.size != n) throw std::runtime_error("search_batch_vl: turns size must match n_envs");
                 if (K < 1) throw std::runtime_error("search_batch_vl: K must be >= 1");
                 if (bi.size != n * BM::board_size()) throw std::runtime_error("search_batch_vl: bad board shape");
                 const py::ssize_t tot = n * K;
                 py::array_t<int8_t> ob(leaf_shape(tot, BM::board_shape()));
                 py::array_t<float> d(tot), p1(tot), p2(tot);
                 py::array_t<uint8_t> it(tot);
                 py::array_t<int> ot(tot), sy(tot);
                 py::array_t<uint8_t> vm(std::vector<py::ssize_t>{tot, BM::action_size()});
                 int rc;
                 {
                     auto *pin = static_cast<const int8_t *>(bi.ptr);
                     auto *ptn = static_cast<const int32_t *>(bt.ptr);
                     auto *pob = ob.mutable_data(); auto *pd = d.mutable_data();
                     auto *pp1 = p1.mutable_data(); auto *pp2 = p2.mutable_data();
                     auto *pit = it.mutable_data(); auto *pot = ot.mutable_data();
                     auto *psy = sy.mutable_data(); auto *pvm = vm.mutable_data();
                     py::gil_scoped_release rel;
                     rc = az_mcts_search_batch_vl(self.h(), K, pin, ptn, n, pob, pd, pp1, pp2, pit, pot, psy, pvm);
                 }
                 check(rc);
                 return py::make_tuple(ob, d, p1, p2, it, ot, sy, vm);
             },
             py::arg("K"), py::arg("input_boards"), py::arg("turns"),
             "VL Selection: K sims per tree, returns N*K leaves + sym_ids + valid_mask")
        .def("backprop_batch_vl",
             [](BM &self, int K, carray<float> policy_logits, carray<float> d_vals, carray<float> p1w_vals,
                carray<float> p2w_vals, carray<float> moves_left, carray<uint8_t> is_term, carray<int> sym_ids) {
                 auto bp = policy_logits.request();
                 const py::ssize_t tot = static_cast<py::ssize_t>(self.n()) * K;
                 const py::ssize_t got = bp.ndim ? bp.shape[0] : 0;
                 if (got != tot)
                     throw std::runtime_error("backprop_batch_vl: policy batch (" + std::to_string(got) +
                                              ") != N*K (" + std::to_string(tot) + ")");
                 if (bp.size != tot * BM::action_size()) throw std::runtime_error("backprop_batch_vl: bad policy shape");
                 if (d_vals.size() != tot || p1w_vals.size() != tot || p2w_vals.size() != tot)
                     throw std::runtime_error("backprop_batch_vl: d/p1w/p2w size must be N*K");
                 if (moves_left.size() != tot) throw std::runtime_error("backprop_batch_vl: moves_left size must be N*K");
                 if (is_term.size() != tot) throw std::runtime_error("backprop_batch_vl: is_term size must be N*K");
                 if (sym_ids.size() != tot) throw std::runtime_error("backprop_batch_vl: sym_ids size must be N*K");
                 int rc;
                 {
                     auto *a = policy_logits.data(); auto *b = d_vals.data(); auto *c = p1w_vals.data();
                     auto *e = p2w_vals.data(); auto *f = moves_left.data(); auto *g = is_term.data();
                     auto *s = sym_ids.data();
                     py::gil_scoped_release rel;
                     rc = az_mcts_backprop_batch_vl(self.h(), K, a, b, c, e, f, g, s, tot);
                 }
                 check(rc);
             },
             py::arg("K"), py::arg("policy_logits"), py::arg("d_vals"), py::arg("p1w_vals"),
             py::arg("p2w_vals"), py::arg("moves_left"), py::arg("is_term"), py::arg("sym_ids"),
             "VL Backprop: remove VL then backprop N*K results")
        .def("search",
             [](BM &self, IEval &ev, carray<int8_t> input_boards, carray<int> turns, int n_playout) {
                 auto bi = input_boards.request();
                 const py::ssize_t n = bi.ndim ? bi.shape[0] : 0;
                 if (n != self.n())
                     throw std::runtime_error("search: input_boards batch size (" + std::to_string(n) +
                                              ") must match n_envs (" + std::to_string(self.n()) + ")");
                 if (turns.size() != n) throw std::runtime_error("search: turns size must match batch size");
                 int rc;
                 {
                     auto *pin = static_cast<const int8_t *>(bi.ptr);
                     auto *ptn = turns.data();
                     py::gil_scoped_release rel;
                     rc = ev.device_rng ? az_mcts_search_rollout_dev(self.h(), pin, ptn, n, n_playout)
                                        : az_mcts_search_rollout(self.h(), pin, ptn, n, n_playout);
                 }
                 check(rc);
             },
             py::arg("evaluator"), py::arg("input_boards"), py::arg("turns"), py::arg("n_playout"),
             "Run MCTS search with the engine's rollout evaluator")
        .def("get_all_counts",
             [](BM &self) {
                 std::vector<int> out(static_cast<size_t>(self.n()) * BM::action_size());
                 check(az_mcts_get_all_counts(self.h(), out.data()));
                 return out;   // Python list, as std::vector<int> in the reference
             })
        .def("get_all_root_stats",
             [](BM &self) {
                 py::array_t<float> out({self.n(), 6 + 8 * BM::action_size()});
                 check(az_mcts_get_all_root_stats(self.h(), out.mutable_data()));
                 return out;
             },
             "Returns root node stats: shape (n_envs, 6 + action_size*8)")
        // ---- extensions (not in the reference): access for the fused device loop ----
        .def_property_readonly("handle", [](BM &self) { return reinterpret_cast<uintptr_t>(self.h()); },
                               "az_mcts* of the C ABI, for the device entry points")
        .def_property_readonly_static("action_size", [](py::object) { return BM::action_size(); })
        .def_property_readonly_static("board_size", [](py::object) { return BM::board_size(); })
        .def_property_readonly_static("board_shape", [](py::object) {
            auto s = BM::board_shape();
            py::tuple t(s.size());
            for (size_t i = 0; i < s.size(); ++i) t[i] = s[i];
            return t;
        });
}

}  // namespace

PYBIND11_MODULE(mcts_cpp, m)
{
    m.doc() = "AlphaZero batched MCTS on MI355X (drop-in for the reference's mcts_cpp)";

    py::class_<az_search_config>(m, "SearchConfig")
        .def(py::init([] {
            az_search_config c{1.25f, 19652.0f, 0.3f, 0.25f, 0.4f, 0.0f, 0.2f, 0.0f, 8.0f, 1.0f, 1, 1};
            return c;
        }))
        .def_readwrite("c_init", &az_search_config::c_init)
        .def_readwrite("c_base", &az_search_config::c_base)
        .def_readwrite("dirichlet_alpha", &az_search_config::dirichlet_alpha)
        .def_readwrite("noise_epsilon", &az_search_config::noise_epsilon)
        .def_readwrite("fpu_reduction", &az_search_config::fpu_reduction)
        .def_readwrite("mlh_slope", &az_search_config::mlh_slope)
        .def_readwrite("mlh_cap", &az_search_config::mlh_cap)
        .def_readwrite("score_utility_factor", &az_search_config::score_utility_factor)
        .def_readwrite("score_scale", &az_search_config::score_scale)
        .def_readwrite("value_decay", &az_search_config::value_decay)
        .def_property(
            "use_symmetry", [](const az_search_config &c) { return c.use_symmetry != 0; },
            [](az_search_config &c, bool v) { c.use_symmetry = v ? 1 : 0; })
        .def_readwrite("vl_count", &az_search_config::vl_count);

    register_game<AZ_GAME_CONNECT4>(m, "Connect4");
    register_game<AZ_GAME_OTHELLO>(m, "Othello");
}
