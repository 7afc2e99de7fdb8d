"""Backend-agnostic search scenarios.

Every function here drives an object with the Python surface of the reference's
`mcts_cpp.BatchedMCTS_Connect4` (src/cpp/mcts_bindings.cpp:50-369).  The same code is run
 * against the compiled reference (tests/golden/make_golden.py -> committed .npz fixtures),
 * against the plain-C oracle (tests/test_oracle_golden.py),
 * against the HIP engine (tests/test_hip_parity.py, -m gpu),
so a fixture mismatch always means a behavioural difference, never a harness difference.

The evaluator used for search parity is HashEval: a pure integer hash of the leaf position
whose outputs are exact in fp32 (small integers and one correctly rounded division), so it
can be reproduced bit-for-bit in numpy, in torch on the GPU and inside a HIP kernel.
"""
import numpy as np

A = 7
ROWS, COLS = 6, 7

# search parameters the reference's actor/server use (server.py:44-72,135)
ACTOR_CFG = dict(c_init=1.4, c_base=1000.0, dirichlet_alpha=0.3, noise_epsilon=0.25,
                 fpu_reduction=0.2, mlh_slope=0.1, mlh_cap=0.2, value_decay=1.0,
                 use_symmetry=True, vl_count=1)
# deterministic variant: no RNG is consumed anywhere
DET_CFG = dict(ACTOR_CFG, dirichlet_alpha=0.0, noise_epsilon=0.0, use_symmetry=False)


def apply_cfg(mcts, cfg):
    c = mcts.config
    for k, v in cfg.items():
        setattr(c, k, v)


# ----------------------------------------------------------------------------- numpy Connect4

def np_drop(board, col, turn):
    """Drop a piece for `turn` into `col` of a (6,7) int8 grid (row 5 is the bottom)."""
    for r in range(ROWS - 1, -1, -1):
        if board[r, col] == 0:
            board[r, col] = turn
            return r
    raise ValueError("column full")


def np_valid(board):
    return [c for c in range(COLS) if board[0, c] == 0]


def np_winner(board):
    for p in (1, -1):
        m = (board == p)
        for r in range(ROWS):
            for c in range(COLS):
                if not m[r, c]:
                    continue
                for dr, dc in ((0, 1), (1, 0), (1, 1), (1, -1)):
                    rr, cc = r + 3 * dr, c + 3 * dc
                    if 0 <= rr < ROWS and 0 <= cc < COLS and all(
                            m[r + i * dr, c + i * dc] for i in range(4)):
                        return p
    return 0


def np_done(board):
    return np_winner(board) != 0 or not (board == 0).any()


def np_turn(board):
    return 1 if int((board != 0).sum()) % 2 == 0 else -1


def random_openings(rng, n, max_plies):
    """n legal, non-terminal positions reached by 0..max_plies uniformly random plies."""
    boards = np.zeros((n, ROWS, COLS), np.int8)
    for i in range(n):
        while True:
            b = np.zeros((ROWS, COLS), np.int8)
            turn = 1
            ok = True
            for _ in range(int(rng.integers(0, max_plies + 1))):
                np_drop(b, int(rng.choice(np_valid(b))), turn)
                turn = -turn
                if np_done(b):
                    ok = False
                    break
            if ok:
                boards[i] = b
                break
    turns = np.array([np_turn(b) for b in boards], np.int32)
    return boards, turns


# ----------------------------------------------------------------------------- HashEval

_M64 = (1 << 64) - 1


def boards_to_bitboards(boards):
    """(n,6,7) int8 -> (bb_p1, bb_p2) uint64 in the reference bit layout (Connect4.h:15-29):
    bit = col*7 + (5-row)."""
    n = boards.shape[0]
    bb = np.zeros((2, n), np.uint64)
    for r in range(ROWS):
        for c in range(COLS):
            bit = np.uint64(1 << (c * 7 + (ROWS - 1 - r)))
            bb[0] |= np.where(boards[:, r, c] == 1, bit, np.uint64(0))
            bb[1] |= np.where(boards[:, r, c] == -1, bit, np.uint64(0))
    return bb[0], bb[1]


def hash64(bb0, bb1, turns):
    """splitmix64-style mix of (bb_p1, bb_p2, side to move); uint64 numpy arrays."""
    with np.errstate(over="ignore"):
        x = bb0 * np.uint64(0x9E3779B97F4A7C15)
        x ^= (bb1 + np.uint64(0x7F4A7C159E3779B9)) * np.uint64(0xBF58476D1CE4E5B9)
        x += np.where(turns == 1, np.uint64(0x94D049BB133111EB), np.uint64(0x2545F4914F6CDD1D))
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def hash_eval_from_hash(h):
    """policy[a] = (1 + 4 bits)/16, wdl_rel = three 5-bit weights normalised, ml = 6 bits/2."""
    n = h.shape[0]
    probs = np.empty((n, A), np.float32)
    for a in range(A):
        probs[:, a] = (1 + ((h >> np.uint64(4 * a)) & np.uint64(15))).astype(np.float32) / np.float32(16)
    w = np.stack([1 + ((h >> np.uint64(s)) & np.uint64(31)) for s in (28, 33, 38)], axis=1)
    tot = w.sum(axis=1, keepdims=True).astype(np.float32)
    wdl = w.astype(np.float32) / tot
    ml = ((h >> np.uint64(43)) & np.uint64(63)).astype(np.float32) / np.float32(2)
    return probs, wdl.astype(np.float32), ml


def hash_eval(boards, turns):
    """boards (n,6,7) int8 as the evaluator sees them (possibly mirrored), turns (n,).
    Returns probs (n,7), relative wdl [draw, win, loss] (n,3), moves_left (n,)."""
    bb0, bb1 = boards_to_bitboards(np.asarray(boards))
    return hash_eval_from_hash(hash64(bb0, bb1, np.asarray(turns)))


class HashPV:
    """pv_func with the reference `predict(state, action_mask)` signature
    (Connect4/Network.py:267-288) backed by HashEval.  `state` is the (n,3,6,7) relative
    feature tensor built by MCTS_cpp.py:15-20; board and turn are recovered from it."""
    n_actions = A

    def __init__(self):
        self.calls = []

    def predict(self, state, action_mask=None):
        state = np.asarray(state)
        turns = state[:, 2, 0, 0].astype(np.int32)
        boards = ((state[:, 0] - state[:, 1]) * turns[:, None, None]).astype(np.int8)
        probs, wdl, ml = hash_eval(boards, turns)
        if action_mask is not None:
            probs = probs * np.asarray(action_mask, dtype=np.float32)
        self.calls.append(int(state.shape[0]))
        return probs, wdl, ml.reshape(-1, 1)


def rel_to_abs(wdl_rel, turns):
    """MCTS_cpp.py:23-30"""
    d, w, l = wdl_rel[:, 0], wdl_rel[:, 1], wdl_rel[:, 2]
    return d, np.where(turns == 1, w, l), np.where(turns == 1, l, w)


# ----------------------------------------------------------------------------- playout loop

class C4Game:
    """What the scenario driver needs to know about a game (Connect4 here)."""
    A = 7

    @staticmethod
    def hash_eval(boards, turns):
        return hash_eval(boards, turns)

    @staticmethod
    def bitboards(boards):
        return boards_to_bitboards(boards)

    @staticmethod
    def advance(board, turn, action):
        """Apply `action` in place if the game is still running; returns the new side to move."""
        if not np_done(board) and 0 <= action < COLS and board[0, action] == 0:
            np_drop(board, int(action), int(turn))
            return -turn
        return turn


def playout(mcts, boards, turns, n_playout, K, evaluator=None, log=None, game=C4Game, after_backprop=None):
    """The reference wrapper's loop (MCTS_cpp.py:110-357) at the mcts_cpp level, without
    cache/time budget: K<=1 -> n_playout single sims; K>1 -> one warm-up sim then VL chunks
    of min(K, remaining).  `after_backprop(i)` is called after the i-th iteration's backprop."""
    n = boards.shape[0]
    A = game.A
    if evaluator is None:
        evaluator = game.hash_eval
    it_no = [0]

    def done_one():
        if after_backprop is not None:
            after_backprop(it_no[0])
        it_no[0] += 1

    def one_plain():
        lb, td, t1, t2, it, lt, vm = mcts.search_batch(boards, turns)
        probs = np.zeros((n, A), np.float32)
        d, p1, p2 = td.copy(), t1.copy(), t2.copy()
        ml = np.zeros(n, np.float32)
        nt = ~it.astype(bool)
        if nt.any():
            pr, wdl, m = evaluator(lb[nt], lt[nt])
            probs[nt] = pr * vm[nt].astype(np.float32)
            dd, a1, a2 = rel_to_abs(wdl, lt[nt])
            d[nt], p1[nt], p2[nt], ml[nt] = dd, a1, a2, m
        if log is not None:
            # the ids of a plain search stay inside the native object (pending_sym_ids_); the oracle can show them
            sy = mcts.pending_sym_ids() if hasattr(mcts, "pending_sym_ids") else None
            log.append(dict(kind="plain", is_term=it.copy(), turns=lt.copy(), boards=lb.copy(),
                            mask=vm.copy(), term=np.stack([td, t1, t2], 1), plain_sym=sy))
        mcts.backprop_batch(probs, d, p1, p2, ml, it)
        done_one()

    if K <= 1:
        for _ in range(n_playout):
            one_plain()
        return
    remaining = n_playout
    if remaining > 0:
        one_plain()
        remaining -= 1
    while remaining > 0:
        k = min(K, remaining)
        remaining -= k
        lb, td, t1, t2, it, lt, sy, vm = mcts.search_batch_vl(k, boards, turns)
        tot = n * k
        probs = np.zeros((tot, A), np.float32)
        d, p1, p2 = td.copy(), t1.copy(), t2.copy()
        ml = np.zeros(tot, np.float32)
        nt = ~it.astype(bool)
        if nt.any():
            pr, wdl, m = evaluator(lb[nt], lt[nt])
            probs[nt] = pr * vm[nt].astype(np.float32)
            dd, a1, a2 = rel_to_abs(wdl, lt[nt])
            d[nt], p1[nt], p2[nt], ml[nt] = dd, a1, a2, m
        if log is not None:
            log.append(dict(kind="vl", is_term=it.copy(), turns=lt.copy(), boards=lb.copy(),
                            mask=vm.copy(), sym=sy.copy(), term=np.stack([td, t1, t2], 1)))
        mcts.backprop_batch_vl(k, probs, d, p1, p2, ml, it, sy)
        done_one()


def counts_of(mcts, n, A=7):
    return np.array(mcts.get_all_counts(), np.int32).reshape(n, A)


def play_plies(mcts, boards, turns, n_playout, K, plies, evaluator=None, record_leaves=False, game=C4Game):
    """`plies` rounds of: playout -> record counts/stats -> argmax action -> prune_roots ->
    apply the move.  Finished games stay in the batch with a terminal root (reference
    behaviour in game.py:83-84 until the whole batch ends; quirk 8 of SURVEY appendix A)."""
    boards = boards.copy()
    turns = turns.copy()
    n = boards.shape[0]
    out = dict(counts=[], stats=[], actions=[], sym=[], leaf_sig=[])
    for _ in range(plies):
        log = [] if record_leaves else None
        playout(mcts, boards, turns, n_playout, K, evaluator, log, game)
        c = counts_of(mcts, n, game.A)
        out["counts"].append(c)
        out["stats"].append(np.array(mcts.get_all_root_stats(), np.float32))
        acts = np.argmax(c, axis=1).astype(np.int32)
        out["actions"].append(acts)
        if record_leaves:
            out["sym"].append(np.concatenate([e["sym"] for e in log if e["kind"] == "vl"] or
                                             [np.zeros(0, np.int32)]).astype(np.int8))
            # order-sensitive signature of every leaf returned during this ply
            sig = np.zeros(4, np.int64)
            for e in log:
                bb0, bb1 = game.bitboards(e["boards"])
                h = hash64(bb0, bb1, e["turns"]).astype(np.int64)
                w = np.arange(1, h.size + 1, dtype=np.int64)
                with np.errstate(over="ignore"):
                    sig[0] += (h * w).sum()
                    sig[1] += (e["is_term"].astype(np.int64) * w).sum()
                    sig[2] += (e["mask"].astype(np.int64).sum(1) * w).sum()
                    sig[3] += ((e["term"] @ np.array([1, 2, 3], np.float32)).astype(np.int64) * w).sum()
            out["leaf_sig"].append(sig)
        mcts.prune_roots(acts)
        for i in range(n):
            turns[i] = game.advance(boards[i], int(turns[i]), int(acts[i]))
    res = dict(counts=np.stack(out["counts"]), stats=np.stack(out["stats"]),
               actions=np.stack(out["actions"]), final_boards=boards, final_turns=turns)
    if record_leaves:
        res["sym"] = np.concatenate(out["sym"]) if out["sym"] else np.zeros(0, np.int8)
        res["leaf_sig"] = np.stack(out["leaf_sig"])
    return res


# ----------------------------------------------------------------------------- scenario table

def scenario_inputs(name):
    """Deterministic inputs of every golden search scenario: (cfg, boards, turns, n, K, plies,
    seed).  seed None = no set_seed call."""
    rng = np.random.default_rng(int.from_bytes(name.encode(), "little") % (2 ** 31))
    if name == "g3_det_n50_k1":
        b, t = random_openings(rng, 64, 8)
        return DET_CFG, b, t, 50, 1, 6, None
    if name == "g3_det_n50_k4":
        b, t = random_openings(rng, 64, 8)
        return DET_CFG, b, t, 50, 4, 6, None
    if name == "g3_det_n200_k4":
        b, t = random_openings(rng, 96, 10)
        return DET_CFG, b, t, 200, 4, 8, None
    if name == "g3_det_n203_k8_deep":
        b, t = random_openings(rng, 32, 24)
        return dict(DET_CFG, c_base=1015.0), b, t, 203, 8, 12, None
    if name == "g4_seeded_actor":
        b, t = random_openings(rng, 32, 6)
        return ACTOR_CFG, b, t, 50, 4, 6, 1234
    if name == "g4_seeded_k1":
        b, t = random_openings(rng, 16, 4)
        return dict(ACTOR_CFG, c_base=250.0), b, t, 40, 1, 5, 7
    if name == "g5_eps_no_alpha":     # quirk 2: priors scaled by (1-eps) at the root
        b, t = random_openings(rng, 16, 6)
        return dict(DET_CFG, noise_epsilon=0.25), b, t, 64, 4, 4, None
    if name == "g5_value_decay":
        b, t = random_openings(rng, 16, 6)
        return dict(DET_CFG, value_decay=0.95), b, t, 64, 4, 4, None
    if name == "g5_mlh_off_fpu0":
        b, t = random_openings(rng, 16, 6)
        return dict(DET_CFG, mlh_slope=0.0, fpu_reduction=0.0), b, t, 64, 4, 4, None
    if name == "g5_vl2_fpu04":
        b, t = random_openings(rng, 16, 6)
        return dict(DET_CFG, vl_count=2, fpu_reduction=0.4, c_init=1.25, c_base=19652.0), b, t, 64, 4, 4, None
    if name == "g5_endgames":         # near-full boards: terminal leaves, terminal roots, draws
        b, t = random_openings(rng, 48, 38)
        return DET_CFG, b, t, 48, 4, 8, None
    if name == "g5_p2_fresh_root":    # quirk 1: fresh tree at a P2-to-move position
        b, t = random_openings(rng, 16, 9)
        keep = t == -1
        return DET_CFG, b[keep], t[keep], 32, 4, 3, None
    raise KeyError(name)


SEARCH_SCENARIOS = [
    "g3_det_n50_k1", "g3_det_n50_k4", "g3_det_n200_k4", "g3_det_n203_k8_deep",
    "g4_seeded_actor", "g4_seeded_k1",
    "g5_eps_no_alpha", "g5_value_decay", "g5_mlh_off_fpu0", "g5_vl2_fpu04", "g5_endgames",
    "g5_p2_fresh_root",
]


def run_search_scenario(make_mcts, name):
    cfg, boards, turns, n, K, plies, seed = scenario_inputs(name)
    m = make_mcts(boards.shape[0])
    apply_cfg(m, cfg)
    if seed is not None:
        m.set_seed(seed)
    return play_plies(m, boards, turns, n, K, plies, record_leaves=True)


# ============================================================================= Othello

OT_A = 65
OT_PASS = 64
_DIRS = [(-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1)]


def ot_start():
    b = np.zeros((8, 8), np.int8)
    b[3, 3] = -1; b[3, 4] = 1; b[4, 3] = 1; b[4, 4] = -1     # Othello.h:63-78
    return b


def ot_flips(board, turn, r, c):
    out = []
    for dr, dc in _DIRS:
        line = []
        rr, cc = r + dr, c + dc
        while 0 <= rr < 8 and 0 <= cc < 8 and board[rr, cc] == -turn:
            line.append((rr, cc)); rr += dr; cc += dc
        if line and 0 <= rr < 8 and 0 <= cc < 8 and board[rr, cc] == turn:
            out += line
    return out


def ot_moves(board, turn):
    return [r * 8 + c for r in range(8) for c in range(8)
            if board[r, c] == 0 and ot_flips(board, turn, r, c)]


def ot_over(board):
    return not (board == 0).any() or (not ot_moves(board, 1) and not ot_moves(board, -1))


def ot_play(board, turn, action):
    r, c = divmod(int(action), 8)
    for rr, cc in ot_flips(board, turn, r, c):
        board[rr, cc] = turn
    board[r, c] = turn


def ot_openings(rng, n, max_plies, want_pass=0):
    """n legal, unfinished Othello positions after 0..max_plies random plies (passes when
    forced).  The first `want_pass` of them are positions where the side to move must pass."""
    boards = np.zeros((n, 8, 8), np.int8)
    turns = np.zeros(n, np.int32)
    i = 0
    while i < n:
        b = ot_start(); t = 1
        for _ in range(int(rng.integers(0, max_plies + 1))):
            mv = ot_moves(b, t)
            if mv:
                ot_play(b, t, int(rng.choice(mv)))
            t = -t
            if ot_over(b):
                break
        if ot_over(b):
            continue
        must_pass = not ot_moves(b, t)
        if i < want_pass and not must_pass:
            continue
        boards[i] = b; turns[i] = t
        i += 1
    return boards, turns


def ot_bitboards(boards):
    """(n,8,8) int8 -> (black, white) uint64, bit = row*8 + col (Othello.h:18-23)."""
    flat = np.asarray(boards).reshape(len(boards), 64)
    w = (np.uint64(1) << np.arange(64, dtype=np.uint64))
    bb0 = ((flat == 1).astype(np.uint64) * w).sum(1, dtype=np.uint64)
    bb1 = ((flat == -1).astype(np.uint64) * w).sum(1, dtype=np.uint64)
    return bb0, bb1


def ot_hash_eval(boards, turns):
    """HashEval for Othello: 65 policy values from five re-mixed words (4 bits each), WDL from
    three 5-bit weights, auxiliary utility = 6 bits / 32 - 1 in [-1, 1); all exact in fp32."""
    bb0, bb1 = ot_bitboards(boards)
    h = hash64(bb0, bb1, np.asarray(turns))
    n = h.shape[0]
    probs = np.empty((n, OT_A), np.float32)
    with np.errstate(over="ignore"):
        for k in range(5):
            hk = h + np.uint64(0x9E3779B97F4A7C15) * np.uint64(k + 1)
            hk ^= hk >> np.uint64(29); hk *= np.uint64(0xBF58476D1CE4E5B9); hk ^= hk >> np.uint64(32)
            for j in range(16):
                a = k * 16 + j
                if a < OT_A:
                    probs[:, a] = (1 + ((hk >> np.uint64(4 * j)) & np.uint64(15))).astype(np.float32) / np.float32(16)
    w = np.stack([1 + ((h >> np.uint64(s)) & np.uint64(31)) for s in (28, 33, 38)], axis=1)
    wdl = w.astype(np.float32) / w.sum(axis=1, keepdims=True).astype(np.float32)
    aux = ((h >> np.uint64(43)) & np.uint64(63)).astype(np.float32) / np.float32(32) - np.float32(1)
    return probs, wdl.astype(np.float32), aux


class OthelloHashPV:
    """pv_func for the reference wrapper on Othello (`predict(state, action_mask)` of
    Othello/Network.py): the (n,3,8,8) relative planes back to board + turn, then ot_hash_eval;
    the third output is the auxiliary utility itself (already in [-1, 1))."""
    n_actions = OT_A

    def predict(self, state, action_mask=None):
        state = np.asarray(state)
        turns = state[:, 2, 0, 0].astype(np.int32)
        boards = ((state[:, 0] - state[:, 1]) * turns[:, None, None]).astype(np.int8)
        probs, wdl, aux = ot_hash_eval(boards, turns)
        if action_mask is not None:
            probs = probs * np.asarray(action_mask, dtype=np.float32)
        return probs, wdl, aux.reshape(-1, 1)


class OthelloGame:
    A = OT_A

    @staticmethod
    def hash_eval(boards, turns):
        return ot_hash_eval(boards, turns)

    @staticmethod
    def bitboards(boards):
        return ot_bitboards(boards)

    @staticmethod
    def advance(board, turn, action):
        if ot_over(board):
            return turn
        if action == OT_PASS:
            return -turn if not ot_moves(board, turn) else turn
        r, c = divmod(int(action), 8)
        if board[r, c] == 0 and ot_flips(board, turn, r, c):
            ot_play(board, turn, action)
            return -turn
        return turn


# server defaults for Othello (server.py:44-72): score utility on, MLH ignored
OT_ACTOR_CFG = dict(c_init=1.4, c_base=2000.0, dirichlet_alpha=0.3, noise_epsilon=0.25, fpu_reduction=0.2,
                    mlh_slope=0.1, mlh_cap=0.2, score_utility_factor=0.15, score_scale=8.0,
                    value_decay=1.0, use_symmetry=True, vl_count=1)
OT_DET_CFG = dict(OT_ACTOR_CFG, dirichlet_alpha=0.0, noise_epsilon=0.0, use_symmetry=False)


_OT_SCEN = {
    # name: (cfg, (n_positions, max_plies, want_pass), n_playout, K, plies, seed)
    "ot_det_n60_k4": (dict(OT_DET_CFG, c_base=300.0), (32, 20, 0), 60, 4, 6, None),
    "ot_det_n100_k1": (dict(OT_DET_CFG, c_base=500.0), (16, 30, 0), 100, 1, 4, None),
    "ot_seeded_actor": (dict(OT_ACTOR_CFG, c_base=250.0), (24, 16, 0), 50, 4, 5, 4321),
    # late positions: forced passes, double-pass terminals, full boards
    "ot_endgames_passes": (dict(OT_DET_CFG, c_base=240.0), (40, 58, 6), 48, 4, 8, None),
    "ot_no_score_utility_decay": (dict(OT_DET_CFG, score_utility_factor=0.0, value_decay=0.96, vl_count=2,
                                       c_base=320.0), (16, 24, 0), 64, 4, 4, None),
}


def _othello_scenario_params(name):
    cfg, _, n, K, plies, seed = _OT_SCEN[name]
    return cfg, None, None, n, K, plies, seed


def othello_scenario_inputs(name):
    cfg, (npos, max_plies, want_pass), n, K, plies, seed = _OT_SCEN[name]
    rng = np.random.default_rng(int.from_bytes(name.encode(), "little") % (2 ** 31))
    b, t = ot_openings(rng, npos, max_plies, want_pass)
    return cfg, b, t, n, K, plies, seed


othello_scenario_inputs.__wrapped__ = _othello_scenario_params


OTHELLO_SCENARIOS = ["ot_det_n60_k4", "ot_det_n100_k1", "ot_seeded_actor", "ot_endgames_passes",
                     "ot_no_score_utility_decay"]


def run_othello_scenario(make_mcts, name, inputs=None):
    """inputs = (boards, turns) replays stored start positions (the fixtures keep them: finding
    forced-pass positions by random play in Python is slow)."""
    if inputs is None:
        cfg, boards, turns, n, K, plies, seed = othello_scenario_inputs(name)
    else:
        cfg, _, _, n, K, plies, seed = othello_scenario_inputs.__wrapped__(name)
        boards, turns = inputs
    m = make_mcts(boards.shape[0])
    apply_cfg(m, cfg)
    if seed is not None:
        m.set_seed(seed)
    return play_plies(m, boards, turns, n, K, plies, record_leaves=True, game=OthelloGame)
