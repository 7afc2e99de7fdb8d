"""Parity of the HIP engine (through libaz_mcts.so's C ABI) with the reference.

Bit-exact checks: visit counts, chosen actions, symmetry ids, every leaf the engine returns
(order-sensitive signature), and root statistics compared as uint32 bit patterns - against
 (a) the committed golden fixtures produced by the compiled reference, and
 (b) the plain-C oracle run side by side on the same seeded inputs (sizes the oracle
     finishes in seconds), plus size-independent invariants at the full benchmark size
     (8192 trees, n_playout 200, K=4).
All of these need a GPU: run with `pytest -m gpu`.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import scenarios as S
from oracle import oracle as O
from test_oracle_golden import bits, check_g2, check_search, load, replay_g2

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "alphazero-al_amd")


@pytest.fixture(scope="module")
def mcts_cpp():
    sys.path.insert(0, ROOT)
    import torch  # noqa: F401  (before the engine library: one HIP runtime per process)
    import __graft_entry__ as ge
    ge.build()
    if PKG not in sys.path:
        sys.path.insert(0, PKG)
    from src import mcts_cpp as m
    return m


def test_g2_single_calls(mcts_cpp):
    check_g2(*replay_g2(mcts_cpp.BatchedMCTS_Connect4))


@pytest.mark.parametrize("name", S.SEARCH_SCENARIOS)
def test_search_scenarios_vs_reference_fixtures(mcts_cpp, name):
    check_search(name, S.run_search_scenario(mcts_cpp.BatchedMCTS_Connect4, name))


def _side_by_side(mcts_cpp, cfg, B, n, K, plies, max_open, seed, rng_seed):
    rng = np.random.default_rng(rng_seed)
    boards, turns = S.random_openings(rng, B, max_open)
    res = []
    for make in (mcts_cpp.BatchedMCTS_Connect4, O.BatchedMCTS_Connect4):
        m = make(B)
        S.apply_cfg(m, cfg)
        if seed is not None:
            m.set_seed(seed)
        res.append(S.play_plies(m, boards, turns, n, K, plies, record_leaves=True))
    hip, orc = res
    assert np.array_equal(hip["counts"], orc["counts"])
    assert np.array_equal(hip["sym"], orc["sym"])
    assert np.array_equal(hip["leaf_sig"], orc["leaf_sig"])
    assert np.array_equal(bits(hip["stats"]), bits(orc["stats"]))


def test_vs_oracle_actor_config_seeded(mcts_cpp):
    _side_by_side(mcts_cpp, S.ACTOR_CFG, 512, 200, 4, 5, 10, seed=99, rng_seed=1)


def test_vs_oracle_odd_batch_and_k(mcts_cpp):
    # batch not a multiple of 8 trees per wave, K not dividing n-1, vl_count 3
    _side_by_side(mcts_cpp, dict(S.ACTOR_CFG, vl_count=3, c_base=335.0), 77, 67, 5, 6, 20, seed=5, rng_seed=2)


@pytest.mark.parametrize("K", [2, 3, 7, 8])
def test_vs_oracle_backup_lane_group_sizes(mcts_cpp, K):
    """k_backprop_spread gives every (tree, k) its own lane group, K rounded up to 2, 4 or 8 groups per tree:
    every group count, trees that do not fill the last wavefront, paths deeper than the eight lanes of a group
    (400 simulations on few trees), duplicate leaves and terminal leaves (20 plies into the game)."""
    _side_by_side(mcts_cpp, dict(S.ACTOR_CFG, c_base=500.0), 13, 400, K, 3, 20, seed=11 + K, rng_seed=4)


def test_vs_oracle_arena_growth(mcts_cpp):
    # 800 simulations x 7 plies outgrow the initial 4096-record arenas several times
    _side_by_side(mcts_cpp, dict(S.DET_CFG, c_base=4000.0), 12, 800, 4, 7, 4, seed=None, rng_seed=3)


def test_host_path_compacts_at_re_rooting(mcts_cpp):
    """Host entry points (search_batch_vl / backprop_batch_vl / prune_roots), ten plies of 300 simulations with the
    most visited move played: a tree creates ~1600 records per ply, ~16 000 over the game - the reference's pools
    would hold them all (MCTSNode.h:165-181: nothing is reclaimed before a reset) - and keeps most of its statistics
    from ply to ply.  The re-rooting compacts the trees (threshold derived from the room reserved SINCE THE PREVIOUS
    RE-ROOTING; with the last call's reservation alone no tree was ever copied on this path and the arenas doubled
    three times): the arenas end at 8192 records per half, less than what was created, and every result equals the
    oracle's bit for bit."""
    rng = np.random.default_rng(4)
    boards, turns = S.random_openings(rng, 24, 2)
    cfg = dict(S.ACTOR_CFG, c_base=1500.0)
    m = mcts_cpp.BatchedMCTS_Connect4(24)
    o = O.BatchedMCTS_Connect4(24)
    res = []
    for eng in (m, o):
        S.apply_cfg(eng, cfg)
        eng.set_seed(17)
        res.append(S.play_plies(eng, boards, turns, 300, 4, 10, record_leaves=True))
    hip, orc = res
    assert np.array_equal(hip["counts"], orc["counts"]) and np.array_equal(hip["sym"], orc["sym"])
    assert np.array_equal(hip["leaf_sig"], orc["leaf_sig"]) and np.array_equal(bits(hip["stats"]), bits(orc["stats"]))
    lib = C.CDLL(os.path.join(PKG, "lib", "libaz_mcts.so"))
    lib.az_mcts_capacity.argtypes = [C.c_void_p]
    lib.az_mcts_capacity.restype = C.c_int64
    lib.az_mcts_counters.argtypes = [C.c_void_p, C.POINTER(C.c_int64 * 8)]
    cnt = (C.c_int64 * 8)()
    assert lib.az_mcts_counters(m.handle, C.byref(cnt)) == 0
    created_per_tree = cnt[2] / 24 * 4            # expansions x at least four legal moves each (the boards are never that full here)
    cap = lib.az_mcts_capacity(m.handle)
    assert cap <= 8192 and created_per_tree > 1.2 * cap, (cap, created_per_tree)


def test_full_size_invariants(mcts_cpp):
    """BASELINE config 1 size: 8192 trees, n_playout 200, vl_batch 4."""
    B, n, K = 8192, 200, 4
    rng = np.random.default_rng(11)
    boards, turns = S.random_openings(rng, 64, 10)
    boards = np.tile(boards, (B // 64, 1, 1)); turns = np.tile(turns, B // 64)
    m = mcts_cpp.BatchedMCTS_Connect4(B)
    S.apply_cfg(m, S.DET_CFG)
    S.playout(m, boards, turns, n, K)
    st = np.array(m.get_all_root_stats())
    c = S.counts_of(m, B)
    assert (st[:, 0] == n).all() and (c.sum(1) == n - 1).all()
    # identical roots give identical trees: every copy of a position agrees with the first
    assert np.array_equal(c.reshape(B // 64, 64, 7), np.broadcast_to(c[:64], (B // 64, 64, 7)))
    assert np.array_equal(bits(st).reshape(B // 64, 64, -1), np.broadcast_to(bits(st[:64]), (B // 64, 64, st.shape[1])))
    # ... and with the oracle on those 64 positions
    o = O.BatchedMCTS_Connect4(64)
    S.apply_cfg(o, S.DET_CFG)
    S.playout(o, boards[:64], turns[:64], n, K)
    assert np.array_equal(c[:64], S.counts_of(o, 64))
    assert np.array_equal(bits(st[:64]), bits(o.get_all_root_stats()))
    # remove_all_vl: idempotent, restores the pre-selection statistics
    m.search_batch_vl(K, boards, turns)
    m.remove_all_vl(K)
    a = np.array(m.get_all_root_stats())
    m.remove_all_vl(K)
    assert np.array_equal(a, np.array(m.get_all_root_stats())) and np.array_equal(bits(a), bits(st))
    # WDL sums are probabilities: D + P1W + P2W == 1 at the root
    assert np.allclose(st[:, 3] + st[:, 4] + st[:, 5], 1.0, atol=1e-5)


def test_wrapper_g6_on_engine(mcts_cpp):
    from src import MCTS_cpp
    from test_boundary_cpu import G6_CASES, _replay_g6
    for tag, cache, K, seed in G6_CASES:
        _replay_g6(MCTS_cpp.BatchedMCTS, tag, cache, K, seed)


def test_selfplay_harness_g8_on_the_hip_engine(mcts_cpp):
    """Fixture G8 - what the reference's `Game.batch_self_play` + `AlphaZeroPlayer` returned on the
    compiled reference (8 games, actor configuration: noise, symmetry, virtual loss, temperature
    schedule, td_steps 2) - reproduced by the same harness (tests/harness.py, checked against the
    reference's own files in the CPU suite) on OUR wrapper, OUR Env objects and the HIP engine."""
    import harness
    from src import MCTS_cpp
    from src.env_cpp.connect4 import Env
    np.random.seed(11)
    w = MCTS_cpp.BatchedMCTS(8, c_init=1.4, c_base=160, alpha=0.3, n_playout=32, noise_epsilon=0.25, fpu_reduction=0.2,
                             use_symmetry=True, mlh_slope=0.1, mlh_cap=0.2)
    w.seed(21)
    data = harness.batch_self_play(w, S.HashPV(), Env, 8, temperature=1.0, temp_decay_moves=6, temp_endgame=0, td_steps=2,
                                   vl_batch=4)
    harness.check_against_g8(data, load("g8_selfplay"), bits)


def test_error_behaviour(mcts_cpp):
    m = mcts_cpp.BatchedMCTS_Connect4(8)
    b = np.zeros((8, 6, 7), np.int8); t = np.ones(8, np.int32)
    with pytest.raises(RuntimeError, match="must match n_envs"):
        m.search_batch(b[:4], t[:4])
    with pytest.raises(RuntimeError, match="K must be >= 1"):
        m.search_batch_vl(0, b, t)
    with pytest.raises(RuntimeError, match="n_envs"):
        m.prune_roots(np.zeros(3, np.int32))
    with pytest.raises(RuntimeError, match="1D"):
        m.prune_roots(np.zeros((8, 1), np.int32))
    with pytest.raises(RuntimeError, match="N\\*K"):
        m.backprop_batch_vl(2, np.zeros((8, 7), np.float32), *[np.zeros(8, np.float32)] * 4,
                            np.zeros(8, np.uint8), np.zeros(8, np.int32))
    m.reset_env(-1); m.reset_env(99)                     # silently ignored (BatchedMCTS.h:93-99)
    # inputs of other dtypes are force-cast like pybind's forcecast
    out = m.search_batch(b.astype(np.float64), t.astype(np.int64))
    assert out[0].dtype == np.int8 and out[5].dtype == np.int32 and out[6].shape == (8, 7)
    assert isinstance(m.get_all_counts(), list) and m.get_all_root_stats().shape == (8, 62)


def test_raw_c_abi_via_ctypes(mcts_cpp):
    """The same entry points a cgo/JNI/ctypes binding would use, without pybind in between."""
    L = C.CDLL(os.path.join(PKG, "lib", "libaz_mcts.so"))
    L.az_last_error.restype = C.c_char_p
    h = C.c_void_p()
    assert L.az_mcts_create(0, 16, -1, C.byref(h)) == 0, L.az_last_error()
    L.az_mcts_config.restype = C.c_void_p
    rng = np.random.default_rng(4)
    boards, turns = S.random_openings(rng, 16, 6)

    class Raw:
        """mcts_cpp surface over ctypes, just enough for scenarios.playout"""
        def __init__(self):
            cfgp = L.az_mcts_config(h)
            self.config = O.OrcConfig.from_address(cfgp)     # same leading float fields
        def p(self, a): return a.ctypes.data_as(C.c_void_p)
        def search_batch(self, b, t):
            n = 16
            ob = np.empty((n, 6, 7), np.int8); d, p1, p2 = (np.empty(n, np.float32) for _ in range(3))
            it = np.empty(n, np.uint8); ot = np.empty(n, np.int32); vm = np.empty((n, 7), np.uint8)
            assert L.az_mcts_search_batch(h, self.p(b), self.p(t), C.c_int64(n), self.p(ob), self.p(d), self.p(p1),
                                          self.p(p2), self.p(it), self.p(ot), self.p(vm)) == 0
            return ob, d, p1, p2, it, ot, vm
        def backprop_batch(self, pol, d, p1, p2, ml, it):
            assert L.az_mcts_backprop_batch(h, self.p(pol), self.p(d), self.p(p1), self.p(p2), self.p(ml),
                                            self.p(it), C.c_int64(16)) == 0
        def search_batch_vl(self, K, b, t):
            n = 16 * K
            ob = np.empty((n, 6, 7), np.int8); d, p1, p2 = (np.empty(n, np.float32) for _ in range(3))
            it = np.empty(n, np.uint8); ot = np.empty(n, np.int32); sy = np.empty(n, np.int32)
            vm = np.empty((n, 7), np.uint8)
            assert L.az_mcts_search_batch_vl(h, K, self.p(b), self.p(t), C.c_int64(16), self.p(ob), self.p(d),
                                             self.p(p1), self.p(p2), self.p(it), self.p(ot), self.p(sy),
                                             self.p(vm)) == 0
            return ob, d, p1, p2, it, ot, sy, vm
        def backprop_batch_vl(self, K, pol, d, p1, p2, ml, it, sy):
            c = np.ascontiguousarray
            assert L.az_mcts_backprop_batch_vl(h, K, self.p(c(pol)), self.p(c(d)), self.p(c(p1)), self.p(c(p2)),
                                               self.p(c(ml)), self.p(c(it)), self.p(c(sy)), C.c_int64(16 * K)) == 0
        def get_all_counts(self):
            out = np.empty(16 * 7, np.int32)
            assert L.az_mcts_get_all_counts(h, self.p(out)) == 0
            return out.tolist()

    raw = Raw()
    for k, v in S.DET_CFG.items():
        if k not in ("use_symmetry", "vl_count"):
            setattr(raw.config, k, v)
    cfg = C.cast(L.az_mcts_config(h), C.POINTER(C.c_uint8))
    cfg[40] = 0                                            # use_symmetry byte after 10 floats
    S.playout(raw, boards, turns, 60, 4)
    o = O.BatchedMCTS_Connect4(16)
    S.apply_cfg(o, S.DET_CFG)
    S.playout(o, boards, turns, 60, 4)
    assert np.array_equal(S.counts_of(raw, 16), S.counts_of(o, 16))
    assert L.az_mcts_search_batch(h, raw.p(boards), raw.p(turns), C.c_int64(3), *([None] * 7)) == 1
    assert b"n_envs" in L.az_last_error()
    L.az_mcts_destroy(h)


def test_g9_rollout_search_bit_exact(mcts_cpp):
    """a27: `search(RolloutEvaluator_<G>(), ...)` on the HIP engine against the reference's own
    outputs - fixture G9 (120 playouts, 12 positions) and g9_rollout_more (Dirichlet noise drawn
    between the playout moves, a second search on the same trees, Othello with passes): playout
    moves and noise come from the reference's mt19937 stream in env order
    (RolloutEvaluator.h:44-46, MCTS.h:352-358).  Visit counts and root statistics bit for bit."""
    from test_oracle_golden import check_rollout_more
    g = load("g9_rollout")
    m = mcts_cpp.BatchedMCTS_Connect4(12)
    S.apply_cfg(m, dict(c_init=4.0, c_base=500.0, dirichlet_alpha=0.0, noise_epsilon=0.0,
                        fpu_reduction=0.0, use_symmetry=False, mlh_slope=0.0, mlh_cap=0.2, value_decay=1.0))
    m.set_seed(3)
    ev = mcts_cpp.RolloutEvaluator_Connect4()
    assert ev.device_rng is False
    m.search(ev, g["boards"], g["turns"], 120)
    assert np.array_equal(S.counts_of(m, 12), g["counts"])
    assert np.array_equal(bits(np.array(m.get_all_root_stats())), bits(g["stats"]))

    def rollout(m, b, t, n):
        ev = (mcts_cpp.RolloutEvaluator_Othello if m.action_size == 65 else mcts_cpp.RolloutEvaluator_Connect4)()
        m.search(ev, b, t, n)
    check_rollout_more(mcts_cpp.BatchedMCTS_Connect4, mcts_cpp.BatchedMCTS_Othello, rollout)


def test_rollout_search_on_device(mcts_cpp):
    """a27 with `device_rng`: the playouts run on the device as well, moves from the device
    generator, so the comparison with the (bit-exact) oracle is statistical; what is exact:
    the simulation budget, and forced lines."""
    from src import MCTS_cpp
    rng = np.random.default_rng(21)
    boards, turns = S.random_openings(rng, 256, 10)
    # tree 0: P1 to move with three in a row on the bottom -> col 3 wins at once
    boards[0] = 0; boards[0, 5, 0:3] = 1; boards[0, 4, 0:3] = -1; turns[0] = 1
    w = MCTS_cpp.BatchedMCTS(256, c_init=4, c_base=500, alpha=0, n_playout=300, noise_epsilon=0.0,
                             fpu_reduction=0.0, use_symmetry=False)               # player.py:84-88
    w.seed(3)
    w._get_rollout_evaluator().device_rng = True
    w.rollout_playout(boards, turns)
    c = w.get_visits_count()
    st = w.get_root_stats()
    assert (st["root_N"] == 300).all() and (c.sum(1) == 299).all()
    assert c[0].argmax() == 3 and st["P1W"][0, 3] > 0.9999
    assert np.allclose(st["root_D"] + st["root_P1W"] + st["root_P2W"], 1.0, atol=1e-5)
    # against the oracle (its own mt19937 stream): mean root value and the most visited move agree
    o = O.BatchedMCTS_Connect4(256)
    S.apply_cfg(o, dict(c_init=4.0, c_base=500.0, dirichlet_alpha=0.0, noise_epsilon=0.0, fpu_reduction=0.0,
                        use_symmetry=False, mlh_slope=0.0, mlh_cap=0.2, value_decay=1.0))
    o.set_seed(3)
    o.search_rollout(boards, turns, 300)
    so = o.get_all_root_stats()
    assert abs(float(st["root_Q"].mean()) - float(so[:, 1].mean())) < 0.03
    co = S.counts_of(o, 256)
    # two independent 300-playout searches of the reference itself agree on the most visited
    # move in about half of these positions; the visit distributions are close on average
    o2 = O.BatchedMCTS_Connect4(256)
    S.apply_cfg(o2, dict(c_init=4.0, c_base=500.0, dirichlet_alpha=0.0, noise_epsilon=0.0, fpu_reduction=0.0,
                         use_symmetry=False, mlh_slope=0.0, mlh_cap=0.2, value_decay=1.0))
    o2.set_seed(4)
    o2.search_rollout(boards, turns, 300)
    co2 = S.counts_of(o2, 256)
    ref_agree = (co.argmax(1) == co2.argmax(1)).mean()
    ref_dist = np.abs(co / 299.0 - co2 / 299.0).mean()
    assert (c.argmax(1) == co.argmax(1)).mean() > ref_agree - 0.12
    assert np.abs(c / 299.0 - co / 299.0).mean() < ref_dist * 1.3 + 0.01


# ------------------------------------------------------------------ Othello (a32, BASELINE config 4)
@pytest.mark.parametrize("name", S.OTHELLO_SCENARIOS)
def test_othello_scenarios_vs_reference_fixtures(mcts_cpp, name):
    from test_oracle_golden import check_othello
    check_othello(name, mcts_cpp.BatchedMCTS_Othello)


def test_othello_vs_oracle_side_by_side(mcts_cpp):
    """Seeded actor configuration (symmetry ids over {0,2,6,7}, Dirichlet noise over up to 33
    edges, score utility), larger batch than the fixtures."""
    rng = np.random.default_rng(31)
    boards, turns = S.ot_openings(rng, 96, 30)
    res = []
    for make in (mcts_cpp.BatchedMCTS_Othello, O.BatchedMCTS_Othello):
        m = make(96)
        S.apply_cfg(m, dict(S.OT_ACTOR_CFG, c_base=400.0))
        m.set_seed(17)
        res.append(S.play_plies(m, boards, turns, 80, 4, 4, record_leaves=True, game=S.OthelloGame))
    hip, orc = res
    assert np.array_equal(hip["counts"], orc["counts"])
    assert np.array_equal(hip["sym"], orc["sym"])
    assert np.array_equal(hip["leaf_sig"], orc["leaf_sig"])
    assert np.array_equal(bits(hip["stats"]), bits(orc["stats"]))


def test_othello_config4_size_invariants(mcts_cpp):
    """BASELINE config 4 shape: 4096 trees, n_playout 400, K=4 (one ply)."""
    B, n, K = 4096, 400, 4
    boards = np.tile(S.ot_start()[None], (B, 1, 1)); turns = np.ones(B, np.int32)
    m = mcts_cpp.BatchedMCTS_Othello(B)
    S.apply_cfg(m, dict(S.OT_DET_CFG, c_base=2000.0))
    S.playout(m, boards, turns, n, K, game=S.OthelloGame)
    st = np.array(m.get_all_root_stats()); c = S.counts_of(m, B, 65)
    assert st.shape == (B, 526) and (st[:, 0] == n).all() and (c.sum(1) == n - 1).all()
    assert (c == c[0]).all()                       # identical roots, identical trees
    o = O.BatchedMCTS_Othello(1)
    S.apply_cfg(o, dict(S.OT_DET_CFG, c_base=2000.0))
    S.playout(o, boards[:1], turns[:1], n, K, game=S.OthelloGame)
    assert np.array_equal(c[0], S.counts_of(o, 1, 65)[0])
    assert np.array_equal(bits(st[0]), bits(o.get_all_root_stats()[0]))
