"""GPU parity of the DEVICE-RESIDENT loop under the reference's random stream.

The loop bench.py times (`az_mcts_dev_select` / `az_mcts_dev_backprop` / `az_mcts_dev_prune_roots`,
and all of them inside `az_mcts_dev_search`) draws symmetry ids and Dirichlet noise from a device
generator, which can never reproduce the reference's mt19937 stream (BatchedMCTS.h:68-84).  Here
the oracle - pinned to the compiled reference on the seeded fixtures G4 - runs the actor's
configuration (alpha 0.3, eps 0.25, symmetry on) and RECORDS what it drew: the symmetry id of
every leaf of every selection call (BatchedMCTS.h:148-154,261-267) and the noise row of every
root expansion and every re-rooting (MCTS.h:113-132,352-358).  `az_mcts_dev_replay` plays that
tape back into the device loop, which must then produce the oracle's visit counts and root
statistics bit for bit (uint32 compare): one differing bit in the mirrored policy gather, in the
FUSED noise path of the batched backup or in the re-rooting kernel fails these tests.
"""
import os
import sys

import numpy as np
import pytest

import scenarios as S
from oracle import oracle as O
from test_oracle_golden import bits

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "alphazero-al_amd")


@pytest.fixture(scope="module")
def env():
    sys.path.insert(0, ROOT)
    import torch  # noqa: F401  (before the engine library: one HIP runtime per process)
    import __graft_entry__ as ge
    ge.build()
    if PKG not in sys.path:
        sys.path.insert(0, PKG)
    import torch
    from src import MCTS_cpp, fused, hash_eval
    return dict(torch=torch, W=MCTS_cpp, F=fused, H=hash_eval)


def _root_edges(game, board, turn):
    """Legal actions of a root position in edge order (ascending), none when the game is over."""
    if game is S.C4Game:
        return [] if S.np_done(board) else S.np_valid(board)
    if S.ot_over(board):
        return []
    mv = S.ot_moves(board, int(turn))
    return mv if mv else [S.OT_PASS]


def _noise_by_edge(game, stats, boards, turns):
    """Root statistics (per ACTION) -> the (B, A) noise rows by EDGE index the kernels store."""
    B, A = boards.shape[0], game.A
    per_action = stats[:, 6:].reshape(B, A, 8)[:, :, 3]
    out = np.zeros((B, A), np.float32)
    for i in range(B):
        ed = _root_edges(game, boards[i], turns[i])
        out[i, :len(ed)] = per_action[i, ed]
    return out


def oracle_with_tape(game, make, cfg, boards, turns, n, K, plies, seed):
    """S.play_plies on the oracle, recording per ply: sym (calls, B*K) int32, the noise rows in force
    after the first iteration (root expansions) and after the re-rooting, counts and stats."""
    B = boards.shape[0]
    boards, turns = boards.copy(), turns.copy()
    m = make(B)
    S.apply_cfg(m, cfg)
    m.set_seed(seed)
    Kw = max(1, K)
    tape = dict(sym=[], noise_search=[], noise_prune=[], counts=[], stats=[], actions=[])
    for _ in range(plies):
        log, first = [], {}

        def hook(i, first=first):
            if i == 0:
                first["noise"] = _noise_by_edge(game, np.array(m.get_all_root_stats(), np.float32), boards, turns)
        S.playout(m, boards, turns, n, K, None, log, game, after_backprop=hook)
        sym = np.zeros((len(log), B * Kw), np.int32)
        for c, e in enumerate(log):
            ids = e["plain_sym"] if e["kind"] == "plain" else e["sym"]
            sym[c, :ids.size] = ids
        tape["sym"].append(sym)
        tape["noise_search"].append(first["noise"])
        c = S.counts_of(m, B, game.A)
        tape["counts"].append(c)
        tape["stats"].append(np.array(m.get_all_root_stats(), np.float32))
        acts = np.argmax(c, axis=1).astype(np.int32)
        tape["actions"].append(acts)
        m.prune_roots(acts)
        for i in range(B):
            turns[i] = game.advance(boards[i], int(turns[i]), int(acts[i]))
        tape["noise_prune"].append(_noise_by_edge(game, np.array(m.get_all_root_stats(), np.float32), boards, turns))
    return tape


def device_loop_with_tape(env, game_name, cfg, boards, turns, n, K, plies, tape, native, follow_tape=True, table_log2=0,
                          table_stats=None):
    """The calls DeviceSelfPlay makes per ply (selfplay.py `step`), fed the oracle's draws."""
    torch, F = env["torch"], env["F"]
    game = S.C4Game if game_name == "Connect4" else S.OthelloGame
    B = boards.shape[0]
    boards, turns = boards.copy(), turns.copy()
    os.environ["AZ_FUSED_GRAPH"] = "0"             # a tape position cannot live in a captured graph
    os.environ["AZ_FUSED_NATIVE"] = "1" if native else "0"
    try:
        w = env["W"].BatchedMCTS(B, c_init=cfg["c_init"], c_base=cfg["c_base"], alpha=cfg["dirichlet_alpha"], n_playout=n,
                                 game_name=game_name, noise_epsilon=cfg["noise_epsilon"], fpu_reduction=cfg["fpu_reduction"],
                                 use_symmetry=cfg["use_symmetry"], mlh_slope=cfg["mlh_slope"], mlh_cap=cfg["mlh_cap"],
                                 value_decay=cfg["value_decay"], score_utility_factor=cfg.get("score_utility_factor", 0.0),
                                 score_scale=cfg.get("score_scale", 8.0))
        w.mcts.config.vl_count = cfg["vl_count"]
        w.seed(99)                                  # the generator's seed must not matter: everything is replayed
        net = (env["H"].HashEvaluator if game_name == "Connect4" else env["H"].OthelloHashEvaluator)("cuda")
        fs = F.FusedSearch(w, net)
        assert (fs._native_model() is not None) == native
        if table_log2:
            fs.enable_table(table_log2)
        L, h = F.lib(), fs.h
        A = game.A
        counts = torch.zeros((B, A), dtype=torch.int32, device="cuda")
        stats = torch.zeros((B, 6 + 8 * A), dtype=torch.float32, device="cuda")
        noise = torch.zeros((B, A), dtype=torch.float32, device="cuda")
        out_c, out_s = [], []
        for p in range(plies):
            sym = torch.from_numpy(tape["sym"][p]).cuda()
            noise.copy_(torch.from_numpy(tape["noise_search"][p]))
            fs.replay(sym, noise)
            fs.upload_roots(boards, turns)
            fs.search(n, K)
            F.check(L.az_mcts_dev_counts(h, counts.data_ptr(), F._stream()))
            F.check(L.az_mcts_dev_root_stats(h, stats.data_ptr(), F._stream()))
            out_c.append(counts.cpu().numpy().copy()); out_s.append(stats.cpu().numpy().copy())
            acts = np.argmax(out_c[-1], axis=1).astype(np.int32)
            if follow_tape:
                assert np.array_equal(acts, tape["actions"][p])
            noise.copy_(torch.from_numpy(tape["noise_prune"][p]))
            a_dev = torch.from_numpy(acts).cuda()
            F.check(L.az_mcts_dev_prune_roots(h, a_dev.data_ptr(), F._stream()))
            torch.cuda.synchronize()
            for i in range(B):
                turns[i] = game.advance(boards[i], int(turns[i]), int(acts[i]))
        fs.replay(None, None)
        if table_stats is not None:
            table_stats.update(fs.table_stats())
        F.check(L.az_mcts_dev_check(h, F._stream()))
        torch.cuda.synchronize()
        F.check(L.az_mcts_dev_check(h, F._stream()))
        return np.stack(out_c), np.stack(out_s)
    finally:
        os.environ.pop("AZ_FUSED_GRAPH", None)
        os.environ.pop("AZ_FUSED_NATIVE", None)


def _compare(tape, counts, stats):
    ref_c, ref_s = np.stack(tape["counts"]), np.stack(tape["stats"])
    assert np.array_equal(counts, ref_c), "visit counts differ from the oracle"
    bad = np.argwhere(bits(stats) != bits(ref_s))
    assert bad.size == 0, "root statistics differ from the oracle at (ply, tree, column) %s" % bad[:5].tolist()
    # the tape was really in force: noise columns are non-trivial and symmetry ids were mixed
    assert (ref_s[:, :, 6:].reshape(ref_s.shape[0], ref_s.shape[1], -1, 8)[..., 3] > 0).any()
    assert len({int(v) for s in tape["sym"] for v in np.unique(s)}) > 1


@pytest.mark.parametrize("native", [True, False], ids=["native_loop", "python_loop"])
@pytest.mark.parametrize("n,K", [(50, 4), (41, 1), (200, 4), (43, 2), (61, 3), (73, 8)])
def test_connect4_device_loop_reference_stream(env, native, n, K):
    """ACTOR_CFG (the configuration bench.py times) with re-rooting between plies; openings up to 30
    plies deep so that terminal leaves (no symmetry draw, BatchedMCTS.h:148) and finished games occur.
    K = 2, 3, 8: the FUSED instantiations of k_backprop_spread for two, four and eight lane groups per tree
    (K = 8 selects with k_select8)."""
    rng = np.random.default_rng(1000 + n + K)
    boards, turns = S.random_openings(rng, 96, 30)
    cfg = dict(S.ACTOR_CFG, c_base=5.0 * n)
    tape = oracle_with_tape(S.C4Game, O.BatchedMCTS_Connect4, cfg, boards, turns, n, K, 5, seed=20 + K)
    counts, stats = device_loop_with_tape(env, "Connect4", cfg, boards, turns, n, K, 5, tape, native)
    _compare(tape, counts, stats)


@pytest.mark.parametrize("native", [True, False], ids=["native_loop", "python_loop"])
@pytest.mark.parametrize("n,K", [(48, 4), (30, 1)])
def test_othello_device_loop_reference_stream(env, native, n, K):
    """Othello: symmetry ids from {0,2,6,7}, up to 33 edges of noise, passes, score utility."""
    rng = np.random.default_rng(77 + n)
    boards, turns = S.ot_openings(rng, 40, 40, 0)
    cfg = dict(S.OT_ACTOR_CFG, c_base=5.0 * n)
    tape = oracle_with_tape(S.OthelloGame, O.BatchedMCTS_Othello, cfg, boards, turns, n, K, 4, seed=5 + K)
    counts, stats = device_loop_with_tape(env, "Othello", cfg, boards, turns, n, K, 4, tape, native)
    _compare(tape, counts, stats)


def test_replay_tape_misuse_is_refused(env):
    """Running past the tape, or a tape narrower than the selection, is an error, not a silent
    fall back to the generator."""
    torch, F = env["torch"], env["F"]
    os.environ["AZ_FUSED_GRAPH"] = "0"
    try:
        w = env["W"].BatchedMCTS(8, c_init=1.4, c_base=100.0, alpha=0.3, n_playout=10)
        fs = F.FusedSearch(w, env["H"].HashEvaluator("cuda"))
        b = np.zeros((8, 6, 7), np.int8); t = np.ones(8, np.int32)
        fs.upload_roots(b, t)
        fs.replay(torch.zeros((3, 8 * 4), dtype=torch.int32, device="cuda"), None)
        with pytest.raises(RuntimeError, match="replay tape"):
            fs.search(10, 4)                    # needs 4 calls, the tape holds 3
        fs.replay(torch.zeros((8, 8), dtype=torch.int32, device="cuda"), None)
        with pytest.raises(RuntimeError, match="replay tape"):
            fs.search(10, 4)                    # K = 4 needs 32 columns
        fs.replay(None, None)
        fs.search(10, 4)
        torch.cuda.synchronize()
    finally:
        os.environ.pop("AZ_FUSED_GRAPH", None)


def test_replay_comparison_is_sensitive(env):
    """The comparison really depends on the tape: one mirrored leaf, or one noise value moved by one
    unit in the last place, changes the device loop's root statistics."""
    rng = np.random.default_rng(4242)
    boards, turns = S.random_openings(rng, 64, 10)
    n, K = 50, 4
    cfg = dict(S.ACTOR_CFG, c_base=5.0 * n)
    tape = oracle_with_tape(S.C4Game, O.BatchedMCTS_Connect4, cfg, boards, turns, n, K, 1, seed=3)
    counts, stats = device_loop_with_tape(env, "Connect4", cfg, boards, turns, n, K, 1, tape, True)
    _compare(tape, counts, stats)
    wrong = {k: [np.array(x, copy=True) for x in v] for k, v in tape.items()}
    wrong["sym"][0][0, :] ^= 1                       # every root shown mirrored in the first call
    _, s2 = device_loop_with_tape(env, "Connect4", cfg, boards, turns, n, K, 1, wrong, True, follow_tape=False)
    assert not np.array_equal(bits(s2), bits(stats))
    wrong = {k: [np.array(x, copy=True) for x in v] for k, v in tape.items()}
    nz = wrong["noise_search"][0] > 0
    wrong["noise_search"][0][nz] = np.nextafter(wrong["noise_search"][0][nz], np.float32(2.0))
    _, s3 = device_loop_with_tape(env, "Connect4", cfg, boards, turns, n, K, 1, wrong, True, follow_tape=False)
    assert not np.array_equal(bits(s3), bits(stats))


# ---------------------------------------------------------------------------- transposition table (f2, BASELINE config 5)

def test_table_on_equals_oracle(env):
    """The device transposition table against the ORACLE (not against the same engine without it):
    actor configuration with symmetry - the key is the mirrored leaf as the evaluator sees it
    (MCTS_cpp.py:150, Cache.py) - replayed draws, hash evaluator, 256 trees of which 64 share a
    position (transpositions across trees), a table small enough for replacement to happen."""
    rng = np.random.default_rng(31)
    boards, turns = S.random_openings(rng, 256, 14)
    boards[:64] = boards[0]; turns[:64] = turns[0]
    n, K = 80, 4
    cfg = dict(S.ACTOR_CFG, c_base=5.0 * n)
    tape = oracle_with_tape(S.C4Game, O.BatchedMCTS_Connect4, cfg, boards, turns, n, K, 4, seed=8)
    for log2 in (12, 18):
        st = {}
        counts, stats = device_loop_with_tape(env, "Connect4", cfg, boards, turns, n, K, 4, tape, True, table_log2=log2, table_stats=st)
        _compare(tape, counts, stats)
        assert st["hits"] > 0 and st["inserts"] > 0 and (log2 != 12 or st["replaced"] > 0), st


def test_othello_table_on_equals_oracle(env):
    """The table for Othello (entries of 65 policy values; key = the leaf under the drawn symmetry of
    {0, 2, 6, 7}, side to move in the order of the two words): native loop, hash evaluator, replayed
    draws, 96 trees of which 32 share a position - bit-exact against the oracle, with hits."""
    rng = np.random.default_rng(123)
    boards, turns = S.ot_openings(rng, 96, 30, 0)
    boards[:32] = boards[0]; turns[:32] = turns[0]
    n, K = 60, 4
    cfg = dict(S.OT_ACTOR_CFG, c_base=5.0 * n)
    tape = oracle_with_tape(S.OthelloGame, O.BatchedMCTS_Othello, cfg, boards, turns, n, K, 3, seed=17)
    for log2 in (10, 16):
        st = {}
        counts, stats = device_loop_with_tape(env, "Othello", cfg, boards, turns, n, K, 3, tape, True, table_log2=log2, table_stats=st)
        _compare(tape, counts, stats)
        assert st["hits"] > 0 and st["inserts"] > 0 and (log2 != 10 or st["replaced"] > 0), st


@pytest.mark.parametrize("games,plies", [(2048, 3), (16384, 1)])
def test_config5_full_size_table_equals_oracle(env, games, plies):
    """BASELINE config 5 at its own sizes - symmetry on, table of 2^20 entries, 2048 games (one
    GPU's share of 16384) and 16384 games on one GPU, n_playout 200, K 4 - through the native loop
    with the table, bit-exact against the oracle on every tree (replayed draws, hash evaluator)."""
    rng = np.random.default_rng(games)
    b, t = S.random_openings(rng, 512, 16)
    boards = np.tile(b, (games // 512, 1, 1)); turns = np.tile(t, games // 512)
    n, K = 200, 4
    cfg = dict(S.ACTOR_CFG, c_base=5.0 * n)
    tape = oracle_with_tape(S.C4Game, O.BatchedMCTS_Connect4, cfg, boards, turns, n, K, plies, seed=games % 1000)
    st = {}
    counts, stats = device_loop_with_tape(env, "Connect4", cfg, boards, turns, n, K, plies, tape, True, table_log2=20, table_stats=st)
    _compare(tape, counts, stats)
    assert (stats[:, :, 0] == n).all() or plies > 1          # every root saw n_playout visits in the first ply
    assert st["hit_rate"] > 0.2, st


def test_config5_network_table_verify_and_refresh(env):
    """With the reference network's HIP twin at config 5's per-GPU size (2048 games, 2^20 entries,
    symmetry on): verify mode evaluates every leaf densely beside the table path and must find no
    differing row; after a weight update `refresh_cache` (MCTS_cpp.py:361-377) re-evaluates the
    resident keys in place - the next search hits them and still finds no differing row, and a
    table that was NOT refreshed is caught by the same check."""
    torch = env["torch"]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_golden import load
    from src import az_net
    wts = load("g7_checkpoint_weights")
    net = az_net.Connect4Net(device="cuda").eval()
    az_net.load_reference_weights(net, {k: wts[k] for k in wts.files})
    rng = np.random.default_rng(5)
    b, t = S.random_openings(rng, 256, 12)
    boards = np.tile(b, (8, 1, 1)); turns = np.tile(t, 8)
    os.environ["AZ_FUSED_GRAPH"] = "0"
    try:
        w = env["W"].BatchedMCTS(2048, 1.4, 1000, 0.3, 200, noise_epsilon=0.25, fpu_reduction=0.2, use_symmetry=True,
                                 mlh_slope=0.1, cache_size=1000000)
        w.seed(11)
        fs = w._fused_runner(net, True)
        assert fs.table_log2 == 20
        fs.enable_table(20, verify=True)
        w.batch_playout(net, boards, turns, vl_batch=4, fused=True)
        st = fs.table_stats()
        assert st["mismatches"] == 0 and st["hit_rate"] > 0.3, st
        inserts0 = st["inserts"]
        # new weights: a stale table must be noticed by the verify pass ...
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(1.03)
        fs.fast = None; fs._fast_version = None                  # snapshot the new weights WITHOUT touching the table
        table_log2, fs.table_log2 = fs.table_log2, 0
        fs._sync_fast_net()
        fs.table_log2 = table_log2
        for i in range(2048):
            w.mcts.reset_env(i)
        w.batch_playout(net, boards, turns, vl_batch=4, fused=True)
        stale = fs.table_stats()
        assert stale["mismatches"] > 0, "the verify pass did not notice outputs of the old weights"
        # ... and refresh_cache repairs it in place: same keys, fresh values, no new inserts needed
        fs.tt_mismatch.zero_()
        w.refresh_cache(net)
        before = fs.table_stats()
        for i in range(2048):
            w.mcts.reset_env(i)
        w.batch_playout(net, boards, turns, vl_batch=4, fused=True)
        after = fs.table_stats()
        assert after["mismatches"] == 0, after
        assert after["hits"] - before["hits"] > 0.5 * (after["lookups"] - before["lookups"]), (before, after)
        assert inserts0 > 0
    finally:
        os.environ.pop("AZ_FUSED_GRAPH", None)


# ---------------------------------------------------------------------------- BASELINE configs 2 and 3/4 at their own sizes

def test_config1_full_size_equals_oracle(env):
    """BASELINE config 1 - the configuration `bench.py` times: Connect4, 8192 games, n_playout 200 (c_base 1000), K 4,
    actor configuration (noise, symmetry) - three plies with a re-rooting between them through the native loop,
    every tree bit-exact against the oracle (replayed draws, hash evaluator)."""
    rng = np.random.default_rng(200)
    b, t = S.random_openings(rng, 512, 18)
    boards = np.tile(b, (16, 1, 1)); turns = np.tile(t, 16)
    n, K = 200, 4
    cfg = dict(S.ACTOR_CFG, c_base=5.0 * n)
    tape = oracle_with_tape(S.C4Game, O.BatchedMCTS_Connect4, cfg, boards, turns, n, K, 3, seed=1)
    counts, stats = device_loop_with_tape(env, "Connect4", cfg, boards, turns, n, K, 3, tape, True)
    _compare(tape, counts, stats)


def test_config2_full_size_equals_oracle(env):
    """BASELINE config 2's per-GPU share - Connect4, 8192 games, n_playout 800 (c_base 4000), K 4, actor
    configuration - one ply through the native loop, every tree bit-exact against the oracle (replayed
    draws, hash evaluator): 6.5 M simulations, trees of ~4000 records, descents up to ~25 levels deep."""
    rng = np.random.default_rng(800)
    b, t = S.random_openings(rng, 512, 20)
    boards = np.tile(b, (16, 1, 1)); turns = np.tile(t, 16)
    n, K = 800, 4
    cfg = dict(S.ACTOR_CFG, c_base=5.0 * n)
    tape = oracle_with_tape(S.C4Game, O.BatchedMCTS_Connect4, cfg, boards, turns, n, K, 1, seed=3)
    counts, stats = device_loop_with_tape(env, "Connect4", cfg, boards, turns, n, K, 1, tape, True)
    _compare(tape, counts, stats)
    assert (stats[:, :, 0] == n).all()


def test_config3_full_size_equals_oracle(env):
    """BASELINE config 3/4 - Othello, 4096 games, n_playout 400 (c_base 2000), K 4, score utility 0.15, noise and
    the four symmetries on - one ply through the native loop, every tree bit-exact against the oracle."""
    rng = np.random.default_rng(400)
    b, t = S.ot_openings(rng, 256, 30, 0)
    boards = np.tile(b, (16, 1, 1)); turns = np.tile(t, 16)
    n, K = 400, 4
    cfg = dict(S.OT_ACTOR_CFG, c_base=5.0 * n)
    tape = oracle_with_tape(S.OthelloGame, O.BatchedMCTS_Othello, cfg, boards, turns, n, K, 1, seed=3)
    counts, stats = device_loop_with_tape(env, "Othello", cfg, boards, turns, n, K, 1, tape, True)
    _compare(tape, counts, stats)
    assert (stats[:, :, 0] == n).all()
