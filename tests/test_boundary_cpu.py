"""CPU-side checks of the drop-in boundary (no GPU needed):

* libaz_mcts.so loads and exports every function include/az_mcts.h declares;
* the CPython modules import under the reference's names and fail LOUDLY without a GPU;
* env_cpp.connect4.Env against golden set G1 (compiled reference);
* the engine's host generator against libstdc++ (rng_std.npz);
* host logic of src/MCTS_cpp.py (wrapper loop, LRU transposition path) against golden set G6,
  with the plain-C oracle monkeypatched in as the native backend - test-only, the product
  has no such switch;
* the reference's own unmodified player.py / game.py running on top of our modules (G8),
  only where /root/reference exists (it never travels to the GPU box).
"""
import ctypes as C
import os
import pickle
import re
import sys
import types

import numpy as np
import pytest

import scenarios as S
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "alphazero-al_amd")
G = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def built():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.build()
    if PKG not in sys.path:
        sys.path.insert(0, PKG)
    return True


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_c_abi_exports_every_declared_symbol(built):
    names = set()
    inc = os.path.join(ROOT, "include")
    for fn in sorted(os.listdir(inc)):
        if fn.endswith(".h"):
            hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(inc, fn)).read(), flags=re.S)
            names |= set(re.findall(r"\b(az_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 45
    lib = C.CDLL(os.path.join(PKG, "lib", "libaz_mcts.so"))
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_modules_import_under_reference_names(built):
    from src import mcts_cpp
    from src.env_cpp.connect4 import Env          # noqa: F401  (reference import path)
    cls = mcts_cpp.BatchedMCTS_Connect4
    assert (cls.action_size, cls.board_size, tuple(cls.board_shape)) == (7, 42, (6, 7))
    cfg = mcts_cpp.SearchConfig()
    got = [cfg.c_init, cfg.c_base, cfg.dirichlet_alpha, cfg.noise_epsilon, cfg.fpu_reduction,
           cfg.mlh_slope, cfg.mlh_cap, cfg.score_utility_factor, cfg.score_scale, cfg.value_decay,
           cfg.use_symmetry, cfg.vl_count]
    ref = [1.25, 19652.0, 0.3, 0.25, 0.4, 0.0, 0.2, 0.0, 8.0, 1.0, True, 1]   # MCTSNode.h:49-60
    assert np.allclose(got, ref)
    assert hasattr(mcts_cpp, "RolloutEvaluator_Connect4") and hasattr(mcts_cpp, "IEvaluator_Connect4")


@pytest.mark.skipif(have_gpu(), reason="checks the no-GPU failure mode")
def test_engine_fails_loudly_without_gpu(built):
    from src import mcts_cpp
    with pytest.raises(RuntimeError, match="GPU"):
        mcts_cpp.BatchedMCTS_Connect4(4)


def test_evaluator_model_object_host_side(built):
    """az_nn_model_* (include/az_nn.h) needs no device to be created: it is a bundle of pointers.
    Creation validates its description, the scratch size is two bf16 activation tensors, and the
    ctypes mirror of the struct (fast_net.ModelWeights) has the C layout (a wrong layout would put
    n_blocks or eps where a pointer is expected and creation would refuse it or accept garbage)."""
    from src.fast_net import ModelWeights, HeadsWeights, glue, MAX_BLOCKS
    L = glue()
    assert L is not None
    fake = 0x1000                                     # any non-null address: nothing is dereferenced here
    w = ModelWeights()
    for n in ("emb_own", "emb_opp", "pos", "stem_w", "stem_b", "pre_w", "qkvg_w", "qn_w", "kn_w", "o_w"):
        setattr(w, n, fake)
    for n in HeadsWeights._PTRS:
        setattr(w.heads, n, fake)
    w.n_blocks = 3
    for i in range(3):
        for f in ("block_w", "block_b", "block_gamma", "block_beta"):
            getattr(w, f)[i] = fake
    w.eps = 1e-5
    h = C.c_void_p()
    assert L.az_nn_model_create(C.byref(w), C.byref(h)) == 0 and h.value
    assert L.az_nn_model_scratch_bytes(h, 32768) == 2 * 32768 * 42 * 64 * 2
    assert L.az_nn_model_scratch_bytes(h, 0) == 0
    # forward refuses bad arguments before touching the device: no scratch, negative batch, rows without n_rows
    assert L.az_nn_model_forward(h, fake, fake, fake, fake, fake, 8, None, None, None, 0, None) == 1
    assert L.az_nn_model_forward(h, fake, fake, fake, fake, fake, -1, None, None, fake, 1 << 30, None) == 1
    assert L.az_nn_model_forward(h, fake, fake, fake, fake, fake, 8, fake, None, fake, 1 << 30, None) == 1
    assert L.az_nn_model_forward(h, fake, fake, fake, fake, fake, 0, None, None, None, 0, None) == 0      # empty batch: nothing to do
    L.az_nn_model_destroy(h)
    bad = ModelWeights.from_buffer_copy(w)
    bad.n_blocks = MAX_BLOCKS + 1
    assert L.az_nn_model_create(C.byref(bad), C.byref(h)) == 1
    bad = ModelWeights.from_buffer_copy(w)
    bad.block_gamma[2] = None
    assert L.az_nn_model_create(C.byref(bad), C.byref(h)) == 1
    bad = ModelWeights.from_buffer_copy(w)
    bad.o_w = None
    assert L.az_nn_model_create(C.byref(bad), C.byref(h)) == 1
    assert L.az_nn_model_create(None, C.byref(h)) == 1
    # the Othello convolution entry point refuses what it does not implement, without a device
    L.az_nn_othello_conv.argtypes = [C.c_void_p] * 8 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    assert L.az_nn_othello_conv(fake, fake, None, None, fake, fake, None, fake, 4, 64, 8, 1, 1, None, None) == 1
    assert L.az_nn_othello_conv(fake, fake, None, None, fake, fake, None, fake, 0, 256, 10, 1, 1, None, None) == 1
    assert L.az_nn_othello_conv(fake, fake, fake, None, fake, fake, None, fake, 4, 256, 10, 1, 1, None, None) == 1


def test_host_generator_matches_libstdcxx(built):
    g = load("rng_std")
    lib = C.CDLL(os.path.join(PKG, "lib", "libaz_mcts.so"))
    lib.az_rng_gamma_selftest.argtypes = [C.c_uint32, C.c_float, C.c_int, C.c_void_p]
    # the fixture draws 16 + 1024 + 1024 integers before the gamma groups, so compare through
    # the oracle's generator (itself pinned to the fixture) on fresh seeds instead
    L = O.lib()
    for seed in (0, 1, 1234, 99991):
        for alpha in (0.3, 0.03, 1.0, 2.5):
            mt = O.OrcMT()
            L.orc_mt_seed(C.byref(mt), seed)
            want = np.zeros(64, np.float32)
            L.orc_gamma_fill(C.byref(mt), alpha, want.ctypes.data_as(C.c_void_p), 64)
            got = np.zeros(64, np.float32)
            lib.az_rng_gamma_selftest(seed, alpha, 64, got.ctypes.data_as(C.c_void_p))
            assert np.array_equal(bits(got), bits(want)), (seed, alpha)
    # and directly against libstdc++ for the first group of seed 0 / alpha 0.3 is covered by
    # test_oracle_golden.test_rng_matches_libstdcxx (same stream position only there)
    assert g["gamma"].shape == (3, 4, 200, 7)


def test_env_against_g1(built):
    from src.env_cpp.connect4 import Env
    g = load("g1_game_logic")
    e = None
    for i in range(len(g["game"])):
        if i == 0 or g["game"][i] != g["game"][i - 1]:
            e = Env()
        b = e.board
        assert b.dtype == np.float32 and np.array_equal(b.astype(np.int8), g["board"][i])
        assert e.turn == g["turn"][i] and e.winPlayer() == g["winner"][i]
        assert e.check_winner() == g["winner"][i]
        assert e.check_full() == bool(g["full"][i]) and e.done() == bool(g["done"][i])
        vm = e.valid_mask()
        assert isinstance(vm, list) and np.array_equal(np.array(vm, np.uint8), g["mask"][i])
        assert e.valid_move() == [c for c in range(7) if g["mask"][i][c]]
        cs = e.current_state()
        assert cs.shape == (1, 3, 6, 7) and cs.dtype == np.float32
        assert np.array_equal(cs[0].astype(np.int8), g["state"][i])
        m = e.apply_symmetry(1)
        assert np.array_equal(m.board.astype(np.int8), g["mirror"][i])
        assert np.array_equal(m.current_state()[0].astype(np.int8), g["mirror_state"][i])
        assert np.array_equal(e.board.astype(np.int8), g["board"][i])      # not in place
        c = pickle.loads(pickle.dumps(e))
        assert c.turn == e.turn and np.array_equal(c.board, e.board)
        if g["action"][i] >= 0:
            e.step(int(g["action"][i]))
    for k in range(len(g["setter_in"])):
        e = Env(g["setter_in"][k].astype(np.float32))
        assert e.turn == g["setter_turn"][k]
        assert np.array_equal(e.board.astype(np.int8), g["setter_board"][k])
        assert np.array_equal(np.array(e.valid_mask(), np.uint8), g["setter_mask"][k])
    assert Env.NUM_SYMMETRIES == 2 and Env.inverse_symmetry_action(1, 2) == 4
    with pytest.raises(RuntimeError):
        Env(np.zeros((5, 7), np.float32))


# ------------------------------------------------------------------ wrapper host logic (G6)

class _OracleBackend(O.BatchedMCTS_Connect4):
    """TEST ONLY: gives the oracle the one extra attribute the wrapper reads."""


def _replay_g6(wrapper_cls, tag, cache, K, seed):
    g = load("g6_wrapper")
    boards, turns = g["boards"].copy(), g["turns"].copy()
    w = wrapper_cls(24, c_init=1.4, c_base=250, alpha=0.3, n_playout=50, game_name="Connect4",
                    cache_size=cache, noise_epsilon=0.25, fpu_reduction=0.2, use_symmetry=True,
                    mlh_slope=0.1, mlh_cap=0.2)
    w.seed(seed)
    pv = S.HashPV()
    cs, ss = [], []
    for _ in range(3):
        w.batch_playout(pv, boards, turns, vl_batch=K)
        c = w.get_visits_count()
        cs.append(c.astype(np.int32))
        ss.append(np.array(w.mcts.get_all_root_stats()))
        acts = np.argmax(c, 1).astype(np.int32)
        w.prune_roots(acts)
        for i in range(24):
            if not S.np_done(boards[i]) and boards[i][0, acts[i]] == 0:
                S.np_drop(boards[i], int(acts[i]), int(turns[i]))
                turns[i] = -turns[i]
    assert np.array_equal(np.stack(cs), g[f"{tag}_counts"]), tag
    assert np.array_equal(bits(np.stack(ss)), bits(g[f"{tag}_stats"])), tag
    assert np.array_equal(np.array(pv.calls, np.int32), g[f"{tag}_calls"]), tag
    if cache:
        assert len(w.cache) == int(g[f"{tag}_cache_len"][0])
    return w


G6_CASES = [("nocache_k4", 0, 4, 5), ("cache_k4", 4096, 4, 5), ("nocache_k1", 0, 1, 9),
            ("cache_k1", 64, 1, 9)]


@pytest.mark.parametrize("tag,cache,K,seed", G6_CASES)
def test_wrapper_host_logic_g6(built, monkeypatch, tag, cache, K, seed):
    from src import MCTS_cpp
    monkeypatch.setitem(MCTS_cpp._BACKENDS, "Connect4", _OracleBackend)
    w = _replay_g6(MCTS_cpp.BatchedMCTS, tag, cache, K, seed)
    d = w.get_root_stats()
    assert d["root_N"].shape == (24,) and d["prior"].shape == (24, 7)
    assert w.run.__func__ is w.batch_playout.__func__


def test_wrapper_vl_cleanup_on_evaluator_failure(built, monkeypatch):
    """MCTS_cpp.py:351-355: an exception inside the evaluator must leave no in-flight visits."""
    from src import MCTS_cpp
    monkeypatch.setitem(MCTS_cpp._BACKENDS, "Connect4", _OracleBackend)
    rng = np.random.default_rng(3)
    boards, turns = S.random_openings(rng, 8, 6)
    w = MCTS_cpp.BatchedMCTS(8, 1.4, 100, 0.0, 20, noise_epsilon=0.0, use_symmetry=False)
    good = S.HashPV()
    w.batch_playout(good, boards, turns, vl_batch=4)
    before = np.array(w.mcts.get_all_root_stats())

    class Boom:
        def predict(self, *a, **k):
            raise ValueError("evaluator died")
    with pytest.raises(ValueError):
        w.batch_playout(Boom(), boards, turns, n_playout=0 + 5, vl_batch=4)
    # the failed call's warm-up never completed; VL of the aborted batch was removed
    w2 = MCTS_cpp.BatchedMCTS(8, 1.4, 100, 0.0, 20, noise_epsilon=0.0, use_symmetry=False)
    w2.batch_playout(S.HashPV(), boards, turns, vl_batch=4)
    a = w.mcts.search_batch(boards, turns)
    b = w2.mcts.search_batch(boards, turns)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert np.array_equal(before, np.array(w2.mcts.get_all_root_stats()))


# ------------------------------------------------------------------ reference callers on our modules (G8)

@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference checkout not present")
def test_reference_selfplay_harness_g8(built, monkeypatch):
    """Game.batch_self_play + AlphaZeroPlayer (reference python, unmodified, imported from
    /root/reference) on top of OUR src.MCTS_cpp / src.env_cpp; native backend = oracle here."""
    numba = types.ModuleType("numba")
    numba.njit = lambda *a, **k: (lambda f: f)
    monkeypatch.setitem(sys.modules, "numba", numba)
    monkeypatch.syspath_prepend(REF)
    monkeypatch.syspath_prepend(PKG)          # ours first: shadows MCTS_cpp / mcts_cpp / env_cpp
    from src import MCTS_cpp
    assert MCTS_cpp.__file__.startswith(PKG)
    monkeypatch.setitem(MCTS_cpp._BACKENDS, "Connect4", _OracleBackend)
    from src.env_cpp.connect4 import Env
    from src.game import Game
    from src.player import AlphaZeroPlayer
    import src.player as ref_player
    assert ref_player.__file__.startswith(REF)

    g = load("g8_selfplay")
    pv = S.HashPV()
    np.random.seed(11)
    player = AlphaZeroPlayer(pv, n_envs=8, c_init=1.4, c_base=160, n_playout=32, alpha=0.3,
                             is_selfplay=1, noise_epsilon=0.25, fpu_reduction=0.2,
                             use_symmetry=True, mlh_slope=0.1, mlh_cap=0.2, vl_batch=4)
    player.mcts.seed(21)
    data = Game(Env()).batch_self_play(player, 8, temperature=1.0, temp_decay_moves=6,
                                       temp_endgame=0, td_steps=2)
    for i, (winner, play) in enumerate(data):
        assert winner == int(g[f"g{i}_winner"][0])
        for j, nm in enumerate(("state", "prob", "z", "steps", "aux", "root_wdl", "mask", "fut")):
            got = np.array([np.asarray(t[j]) for t in play])
            ref = g[f"g{i}_{nm}"]
            assert got.shape == ref.shape, (i, nm)
            if ref.dtype.kind == "f":
                assert np.array_equal(bits(got), bits(ref)), (i, nm)
            else:
                assert np.array_equal(got, ref), (i, nm)


def test_restated_selfplay_harness_g8(built, monkeypatch):
    """tests/harness.py - the restatement of Game.batch_self_play + get_batch_action that the GPU suite
    runs on the HIP engine (the reference's files do not exist on the GPU box) - reproduces fixture G8
    on our wrapper and Env objects, the oracle standing in for the native backend."""
    import harness
    from src import MCTS_cpp
    monkeypatch.setitem(MCTS_cpp._BACKENDS, "Connect4", _OracleBackend)
    from src.env_cpp.connect4 import Env
    np.random.seed(11)
    w = MCTS_cpp.BatchedMCTS(8, c_init=1.4, c_base=160, alpha=0.3, n_playout=32, noise_epsilon=0.25, fpu_reduction=0.2,
                             use_symmetry=True, mlh_slope=0.1, mlh_cap=0.2)
    w.seed(21)
    data = harness.batch_self_play(w, S.HashPV(), Env, 8, temperature=1.0, temp_decay_moves=6, temp_endgame=0, td_steps=2,
                                   vl_batch=4)
    harness.check_against_g8(data, load("g8_selfplay"), bits)


def test_othello_env_against_reference_fixture(built):
    from src.env_cpp.othello import Env
    from test_oracle_golden import replay_othello_env
    g = replay_othello_env(Env)
    inv = np.array([[Env.inverse_symmetry_action(sid, a) for a in range(65)] for sid in range(8)], np.int32)
    assert np.array_equal(inv, g["inv_action"])
    e = Env()
    assert Env.NUM_SYMMETRIES == 8 and e.board.dtype == np.float32 and e.current_state().shape == (1, 3, 8, 8)
    c = pickle.loads(pickle.dumps(e))
    assert c.turn == e.turn and np.array_equal(c.board, e.board)
    from src import mcts_cpp
    cls = mcts_cpp.BatchedMCTS_Othello
    assert (cls.action_size, cls.board_size, tuple(cls.board_shape)) == (65, 64, (8, 8))
    assert hasattr(mcts_cpp, "RolloutEvaluator_Othello")


def test_gomoku_env_against_reference_fixture(built):
    """`src.env_cpp.gomoku.Env` (surface only - the reference binds no Gomoku search): random
    games on six board / win-length configurations replayed against what the reference's
    compiled Env produced (fixture g1_gomoku_logic): boards, side to move, result, NN planes,
    the eight symmetries, the action map, re-import through the board setter, pickling and the
    error messages of Gomoku.h."""
    from src.env_cpp.gomoku import Env
    g = load("g1_gomoku_logic")
    keys = sorted({k.rsplit("_cfg", 1)[0] for k in g.files if k.endswith("_cfg")})
    assert len(keys) == 18
    for k in keys:
        size, need = (int(v) for v in g[k + "_cfg"])
        e = Env(size, need)
        assert (e.board_size, e.rows, e.cols, e.n_in_row, e.action_size, e.num_symmetries) == (size, size, size, need, size * size, 8)
        acts = g[k + "_actions"]
        for t, a in enumerate(acts):
            assert not e.done() and e.valid_mask()[a] and a in e.valid_move()
            if t % 2:
                e.step(int(a))
            else:
                e.step_xy(*e.action_to_coord(int(a)))
            assert np.array_equal(np.asarray(e.board), g[k + "_boards"][t].astype(np.float32)) and e.board.dtype == np.float32
            assert (e.turn, e.winPlayer(), e.check_winner(), int(e.done())) == (
                int(g[k + "_turns"][t]), int(g[k + "_winners"][t]), int(g[k + "_winners"][t]), int(g[k + "_dones"][t]))
            st = e.current_state()
            assert st.shape == (1, 3, size, size) and st.dtype == np.float32
            assert np.array_equal(st[0].astype(np.int8), g[k + "_states"][t])
        with pytest.raises(RuntimeError, match="game is already finished"):
            e.step(0)
        mid = Env(size, need)
        for a in acts[: max(1, len(acts) // 2)]:
            mid.step(int(a))
        for sid in range(8):
            assert np.array_equal(np.asarray(mid.apply_symmetry(sid).board).astype(np.int8), g[k + "_sym_boards"][sid])
            assert [mid.inverse_symmetry_action(sid, a) for a in range(size * size)] == list(g[k + "_sym_actions"][sid])
        c = mid.copy()
        c.apply_symmetry(3, inplace=True)
        assert np.array_equal(np.asarray(c.board).astype(np.int8), g[k + "_sym_boards"][3])
        imp = Env(g[k + "_boards"][-1].astype(np.float32), need)
        assert [imp.turn, imp.winPlayer(), int(imp.done()), int(imp.check_full())] == list(g[k + "_import"])
        p = pickle.loads(pickle.dumps(mid))
        assert p.turn == mid.turn and p.n_in_row == need and np.array_equal(p.board, mid.board)
    e = Env()
    assert (e.board_size, e.n_in_row, Env.NUM_SYMMETRIES) == (15, 5, 8)
    for bad, msg in (((0, 5), "board_size must be positive"), ((5, 1), "n_in_row must be >= 2"), ((4, 5), "n_in_row must be <= board size")):
        with pytest.raises(RuntimeError, match=msg):
            Env(*bad)
    e.step(7)
    for fn, msg in ((lambda: e.step(7), "cell is already occupied"), (lambda: e.step(225), "action out of range"),
                    (lambda: e.coord_to_action(15, 0), "row/col out of range"), (lambda: e.apply_symmetry(8), "invalid symmetry id")):
        with pytest.raises(RuntimeError, match=msg):
            fn()
    with pytest.raises(RuntimeError, match="turn must be 1 or -1"):
        e.turn = 0
    with pytest.raises(RuntimeError, match="board must be square"):
        Env(np.zeros((3, 4), np.float32))
    e.set_params(6, 4)
    assert e.action_size == 36 and e.turn == 1 and e.valid_mask().count(True) == 36


def test_othello_network_port_matches_reference_outputs_g13(built):
    """az_net.OthelloNet against the reference's Othello CNN (fixture G13: a 32-channel instance
    with trained-looking BatchNorm statistics and non-zero heads, evaluated by the reference on
    the CPU): its state dict loads strictly, forward and predict agree to float rounding."""
    import torch
    from src.az_net import OthelloNet, load_reference_weights
    g, w = load("g13_othello_network"), load("g13_othello_weights")
    net = OthelloNet(h_dim=32, num_res_blocks=2)
    load_reference_weights(net, {k: w[k] for k in w.files})
    planes = g["planes"].astype(np.float32)
    masks = g["masks"].astype(bool)
    with torch.no_grad():
        lp, lv, aux = net(torch.from_numpy(planes), action_mask=torch.from_numpy(masks))
    assert np.abs(lp.numpy() - g["logp"]).max() < 2e-5
    assert np.abs(lv.numpy() - g["logv"]).max() < 2e-5
    assert np.abs(aux.numpy() - g["aux"]).max() < 2e-5
    assert g["logp"].std() > 0.05 and np.abs(g["aux"]).max() > 0.01          # the fixture is not a constant
    p, v, u = net.predict(planes, masks)
    assert p.shape == (len(planes), 65) and u.shape == (len(planes), 1)
    assert np.abs(p - g["probs"]).max() < 1e-5 and np.abs(v - g["wdl"]).max() < 1e-5 and np.abs(u - g["utility"]).max() < 1e-5
    with pytest.raises(ValueError):
        net(torch.from_numpy(planes[:2]))                                     # the mask is an input feature
    fresh = OthelloNet(h_dim=32, num_res_blocks=1)                            # zero heads: uniform outputs
    p, v, u = fresh.predict(planes[:4], masks[:4])
    assert np.allclose(p, 1 / 65, atol=1e-6) and np.allclose(v, 1 / 3, atol=1e-6) and np.allclose(u, 0, atol=1e-7)


def test_othello_weight_packing_is_the_kernels_fragment_order(built):
    """fast_othello.pack_conv_weight: element [tap][k chunk][channel tile][lane][j] must be
    W[16 * tile + lane % 16, 32 * chunk + 8 * (lane // 16) + j, tap // 3, tap % 3] - what lane `lane`
    of a wavefront feeds the MFMA as its A operand (nn_othello.hip)."""
    import torch
    from src.fast_othello import pack_conv_weight
    torch.manual_seed(0)
    for cin in (32, 256):
        w = torch.randn(256, cin, 3, 3)
        wp = pack_conv_weight(w).float()
        assert wp.shape == (9, cin // 32, 16, 4, 16, 8) and wp.is_contiguous()
        wb = w.to(torch.bfloat16).float()
        rng = np.random.default_rng(cin)
        for _ in range(200):
            tap, kc, tile, lane, j = (int(rng.integers(0, n)) for n in (9, cin // 32, 16, 64, 8))
            got = wp[tap, kc, tile, lane // 16, lane % 16, j]
            assert got == wb[16 * tile + lane % 16, 32 * kc + 8 * (lane // 16) + j, tap // 3, tap % 3]
