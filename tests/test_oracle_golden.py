"""Pin the plain-C oracle (oracle/*.c) to the compiled reference's outputs.

Fixtures in tests/golden/ were produced by tests/golden/make_golden.py from the real
reference (oracle/_ref/native, OMP_NUM_THREADS=1).  Everything here is bit-exact: integer
outputs by array_equal, float outputs by comparing their uint32 bit patterns.
CPU only (no GPU marker).
"""
import ctypes as C
import os

import numpy as np
import pytest

import scenarios as S
from oracle import oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


# ------------------------------------------------------------------ libstdc++ <random> restatement
def test_rng_matches_libstdcxx():
    g = load("rng_std")
    L = O.lib()
    for si, seed in enumerate(g["seeds"]):
        mt = O.OrcMT()
        L.orc_mt_seed(C.byref(mt), int(seed))
        assert [L.orc_mt_next(C.byref(mt)) for _ in range(16)] == g["mt"][si].tolist()
        assert [L.orc_uniform_int(C.byref(mt), 0, 1) for _ in range(1024)] == g["bits"][si].tolist()
        assert [L.orc_uniform_int(C.byref(mt), 0, i % 7) for i in range(1024)] == g["ranged"][si].tolist()
        for ai, alpha in enumerate(g["alphas"]):
            for grp in range(200):
                cnt = 1 + grp % 7
                out = np.zeros(7, np.float32)
                L.orc_gamma_fill(C.byref(mt), float(alpha), out.ctypes.data_as(C.c_void_p), cnt)
                assert np.array_equal(bits(out), bits(g["gamma"][si, ai, grp])), (seed, alpha, grp)
        assert L.orc_mt_next(C.byref(mt)) == int(g["tail"][si])


# ------------------------------------------------------------------ G1 game logic
def test_g1_game_logic():
    g = load("g1_game_logic")
    e = None
    for i in range(len(g["game"])):
        if i == 0 or g["game"][i] != g["game"][i - 1]:
            e = O.Connect4Env()
        assert np.array_equal(e.board.astype(np.int8), g["board"][i])
        assert e.turn == g["turn"][i]
        assert e.winPlayer() == g["winner"][i]
        assert e.check_full() == bool(g["full"][i])
        assert e.done() == bool(g["done"][i])
        assert np.array_equal(np.array(e.valid_mask(), np.uint8), g["mask"][i])
        assert np.array_equal(e.current_state()[0].astype(np.int8), g["state"][i])
        m = e.apply_symmetry(1)
        assert np.array_equal(m.board.astype(np.int8), g["mirror"][i])
        assert np.array_equal(m.current_state()[0].astype(np.int8), g["mirror_state"][i])
        # import of the exported grid reproduces the same bitboards (Connect4.h:87-129)
        r = O.Connect4Env(g["board"][i].astype(np.float32))
        assert r.bitboards == e.bitboards and r.turn == e.turn
        if g["action"][i] >= 0:
            e.step(int(g["action"][i]))


def test_g1_board_setter():
    g = load("g1_game_logic")
    for k in range(len(g["setter_in"])):
        e = O.Connect4Env(g["setter_in"][k].astype(np.float32))
        assert e.turn == g["setter_turn"][k]
        assert np.array_equal(e.board.astype(np.int8), g["setter_board"][k])
        assert np.array_equal(np.array(e.valid_mask(), np.uint8), g["setter_mask"][k])


# ------------------------------------------------------------------ G2 single calls
def replay_g2(make):
    g = load("g2_single_calls")
    b, t = g["boards"], g["turns"]
    m = make(6)
    S.apply_cfg(m, S.DET_CFG)
    got = {}
    for it in range(6):
        res = m.search_batch(b, t)
        for j, nm in enumerate(("lb", "td", "t1", "t2", "it", "lt", "vm")):
            got[f"s{it}_{nm}"] = res[j]
        lb, td, t1, t2, itm, lt, vm = res
        pr, wdl, ml = S.hash_eval(lb, lt)
        d, p1, p2 = S.rel_to_abs(wdl, lt)
        nt = ~itm.astype(bool)
        probs = np.where(nt[:, None], pr * vm, 0).astype(np.float32)
        m.backprop_batch(probs, np.where(nt, d, td), np.where(nt, p1, t1), np.where(nt, p2, t2),
                         np.where(nt, ml, 0).astype(np.float32), itm)
        got[f"s{it}_counts"] = np.array(m.get_all_counts(), np.int32)
        got[f"s{it}_stats"] = m.get_all_root_stats()
    res = m.search_batch_vl(3, b, t)
    for j, nm in enumerate(("lb", "td", "t1", "t2", "it", "lt", "sy", "vm")):
        got[f"vl_{nm}"] = res[j]
    m.remove_all_vl(3)
    m.remove_all_vl(3)
    got["vl_removed_stats"] = m.get_all_root_stats()
    res2 = m.search_batch(b, t)
    got["after_remove_lb"] = res2[0]
    got["after_remove_it"] = res2[4]
    return g, got


def check_g2(g, got):
    for k, v in got.items():
        ref = g[k]
        v = np.asarray(v)
        assert v.dtype == ref.dtype and v.shape == ref.shape, k
        if v.dtype == np.float32:
            assert np.array_equal(bits(v), bits(ref)), k
        else:
            assert np.array_equal(v, ref), k


def test_g2_single_calls():
    check_g2(*replay_g2(O.BatchedMCTS_Connect4))


# ------------------------------------------------------------------ G3-G5 search scenarios
def check_search(name, got):
    g = load(name)
    assert np.array_equal(got["counts"], g["counts"]), "visit counts"
    assert np.array_equal(got["actions"], g["actions"])
    assert np.array_equal(bits(got["stats"]), bits(g["stats"])), "root stats"
    assert np.array_equal(got["sym"], g["sym"]), "symmetry ids"
    assert np.array_equal(got["leaf_sig"], g["leaf_sig"]), "leaf outputs"
    assert np.array_equal(got["final_boards"], g["final_boards"])


@pytest.mark.parametrize("name", S.SEARCH_SCENARIOS)
def test_search_scenarios(name):
    check_search(name, S.run_search_scenario(O.BatchedMCTS_Connect4, name))


def test_invariants_on_oracle():
    """SURVEY section 4 invariants, checked on the oracle itself."""
    rng = np.random.default_rng(5)
    boards, turns = S.random_openings(rng, 32, 6)
    m = O.BatchedMCTS_Connect4(32)
    S.apply_cfg(m, S.DET_CFG)
    S.playout(m, boards, turns, 50, 4)
    st = m.get_all_root_stats()
    c = S.counts_of(m, 32)
    assert (st[:, 0] == 50).all() and (c.sum(1) == 49).all()
    # remove_all_vl is idempotent and leaves no in-flight visits behind
    m.search_batch_vl(4, boards, turns)
    m.remove_all_vl(4)
    before = m.get_all_root_stats().copy()
    m.remove_all_vl(4)
    assert np.array_equal(before, m.get_all_root_stats())
    ref = O.BatchedMCTS_Connect4(32)
    S.apply_cfg(ref, S.DET_CFG)
    S.playout(ref, boards, turns, 50, 4)
    a = m.search_batch(boards, turns)
    b = ref.search_batch(boards, turns)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


# ------------------------------------------------------------------ G9 rollout search
def test_g9_rollout_search():
    g = load("g9_rollout")
    m = O.BatchedMCTS_Connect4(12)
    S.apply_cfg(m, dict(c_init=4.0, c_base=500.0, dirichlet_alpha=0.0, noise_epsilon=0.0,
                        fpu_reduction=0.0, use_symmetry=False, mlh_slope=0.0, mlh_cap=0.2,
                        value_decay=1.0))
    m.set_seed(3)
    m.search_rollout(g["boards"], g["turns"], 120)
    assert np.array_equal(S.counts_of(m, 12), g["counts"])
    assert np.array_equal(bits(m.get_all_root_stats()), bits(g["stats"]))


ROLLOUT_C4 = dict(c_init=4.0, c_base=500.0, dirichlet_alpha=0.3, noise_epsilon=0.25, fpu_reduction=0.2,
                  use_symmetry=False, mlh_slope=0.0, mlh_cap=0.2, value_decay=1.0)
ROLLOUT_OT = dict(c_init=4.0, c_base=300.0, dirichlet_alpha=0.3, noise_epsilon=0.25, fpu_reduction=0.2,
                  use_symmetry=False, mlh_slope=0.0, mlh_cap=0.2, value_decay=1.0, score_utility_factor=0.15,
                  score_scale=8.0)


def check_rollout_more(make_c4, make_ot, rollout):
    """g9_rollout_more: root noise drawn between the playout moves, a second search on the same
    trees, and Othello rollouts (passes, score utility).  rollout(m, boards, turns, n)."""
    g = load("g9_rollout_more")
    m = make_c4(24)
    S.apply_cfg(m, ROLLOUT_C4)
    m.set_seed(11)
    for tag in ("1", "2"):
        rollout(m, g["c4_boards"], g["c4_turns"], 90)
        assert np.array_equal(S.counts_of(m, 24), g["c4_counts" + tag])
        assert np.array_equal(bits(np.array(m.get_all_root_stats())), bits(g["c4_stats" + tag]))
    m = make_ot(16)
    S.apply_cfg(m, ROLLOUT_OT)
    m.set_seed(5)
    rollout(m, g["ot_boards"], g["ot_turns"], 60)
    assert np.array_equal(S.counts_of(m, 16, 65), g["ot_counts"])
    assert np.array_equal(bits(np.array(m.get_all_root_stats())), bits(g["ot_stats"]))


def test_g9_rollout_more():
    check_rollout_more(O.BatchedMCTS_Connect4, O.BatchedMCTS_Othello, lambda m, b, t, n: m.search_rollout(b, t, n))


# ------------------------------------------------------------------ Othello (a32, config 4)
def check_othello(name, make):
    g = load(name)
    got = S.run_othello_scenario(make, name, inputs=(g["in_boards"], g["in_turns"]))
    assert np.array_equal(got["counts"], g["counts"]), "visit counts"
    assert np.array_equal(got["actions"], g["actions"])
    assert np.array_equal(bits(got["stats"]), bits(g["stats"])), "root stats"
    assert np.array_equal(got["sym"], g["sym"]), "symmetry ids"
    assert np.array_equal(got["leaf_sig"], g["leaf_sig"]), "leaf outputs"
    assert np.array_equal(got["final_boards"], g["final_boards"])


@pytest.mark.parametrize("name", S.OTHELLO_SCENARIOS)
def test_othello_search_scenarios(name):
    check_othello(name, O.BatchedMCTS_Othello)


def replay_othello_env(make_env):
    g = load("g1_othello_logic")
    e = None
    for i in range(len(g["game"])):
        if i == 0 or g["game"][i] != g["game"][i - 1]:
            e = make_env()
        assert np.array_equal(np.asarray(e.board).astype(np.int8), g["board"][i])
        assert e.turn == g["turn"][i] and e.winPlayer() == g["winner"][i]
        assert e.check_full() == bool(g["full"][i]) and e.done() == bool(g["done"][i])
        assert np.array_equal(np.array(e.valid_mask(), np.uint8), g["mask"][i])
        assert np.array_equal(e.current_state()[0].astype(np.int8), g["state"][i])
        for sid in range(8):
            assert np.array_equal(np.asarray(e.apply_symmetry(sid).board).astype(np.int8), g["syms"][i][sid])
        if g["action"][i] >= 0:
            e.step(int(g["action"][i]))
    return g


def test_othello_game_logic():
    replay_othello_env(O.OthelloEnv)
