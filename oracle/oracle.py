"""ctypes front-end of the plain-C oracle (oracle/*.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module, and only as the checker.  The classes mirror the Python surface of the reference's
pybind module (src/cpp/mcts_bindings.cpp:50-369, env_common.h:133-249) so that one test
driver can run the compiled reference, this oracle and the HIP engine side by side.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

A = 7
CELLS = 42
STATS = 6 + 8 * A


class OrcConfig(C.Structure):
    """Field order == oracle.h orc_config == reference SearchConfig (MCTSNode.h:47-61)."""
    _fields_ = [
        ("c_init", C.c_float), ("c_base", C.c_float), ("dirichlet_alpha", C.c_float),
        ("noise_epsilon", C.c_float), ("fpu_reduction", C.c_float), ("mlh_slope", C.c_float),
        ("mlh_cap", C.c_float), ("score_utility_factor", C.c_float), ("score_scale", C.c_float),
        ("value_decay", C.c_float), ("use_symmetry", C.c_int), ("vl_count", C.c_int),
    ]


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in
                ("sims", "levels", "expansions", "terminal_hits", "dup_leaves", "backup_nodes")]


class OrcC4(C.Structure):
    _fields_ = [("cells", C.c_int8 * CELLS), ("turn", C.c_int), ("bb", C.c_uint64 * 2),
                ("height", C.c_int * 7), ("n_pieces", C.c_int), ("last_player", C.c_int)]


class OroOT(C.Structure):
    _fields_ = [("cells", C.c_int8 * 64), ("turn", C.c_int), ("bb", C.c_uint64 * 2),
                ("n_pieces", C.c_int), ("passes", C.c_int), ("last_player", C.c_int)]


class OrcMT(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("idx", C.c_int)]


def build(force=False):
    """Compile liboracle.so with gcc (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("c4_oracle.c", "mcts_oracle.c", "rng_oracle.c", "othello_oracle.c", "mcts_impl.inc", "oracle.h")):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    vp, i32, u8p = C.c_void_p, C.c_int, C.c_void_p
    for pfx in ("orc", "oro"):
        f = lambda n: getattr(L, pfx + "_" + n)   # noqa: E731
        f("create").restype = vp; f("create").argtypes = [i32]
        f("destroy").argtypes = [vp]
        f("config_ptr").restype = C.POINTER(OrcConfig); f("config_ptr").argtypes = [vp]
        f("set_seed").argtypes = [vp, i32]
        f("reset_env").argtypes = [vp, i32]
        f("prune_roots").argtypes = [vp, vp]
        f("search_batch").argtypes = [vp] + [vp] * 9
        f("backprop_batch").argtypes = [vp] + [vp] * 6
        f("remove_all_vl").argtypes = [vp, i32]
        f("search_batch_vl").argtypes = [vp, i32] + [vp] * 10
        f("backprop_batch_vl").argtypes = [vp, i32] + [vp] * 7
        f("search_rollout").argtypes = [vp, vp, vp, i32]
        f("get_all_counts").argtypes = [vp, vp]
        f("get_all_root_stats").argtypes = [vp, vp]
        f("pending_sym").argtypes = [vp, vp]
        f("stats_get").argtypes = [vp, C.POINTER(OrcStats)]
        f("stats_reset").argtypes = [vp]
        f("tree_size").argtypes = [vp, i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    for name in ("oro_ot_reset", "oro_ot_export_cells"):
        getattr(L, name).argtypes = [C.POINTER(OroOT)]
    L.oro_ot_import.argtypes = [C.POINTER(OroOT), vp]
    L.oro_ot_step.argtypes = [C.POINTER(OroOT), i32]
    L.oro_ot_winner.argtypes = [C.POINTER(OroOT)]; L.oro_ot_winner.restype = i32
    L.oro_ot_full.argtypes = [C.POINTER(OroOT)]; L.oro_ot_full.restype = i32
    L.oro_ot_valid_moves.argtypes = [C.POINTER(OroOT), vp]; L.oro_ot_valid_moves.restype = i32
    L.oro_ot_apply_sym.argtypes = [C.POINTER(OroOT), i32]
    L.oro_ot_current_state.argtypes = [C.POINTER(OroOT), vp]
    for name in ("orc_c4_reset", "orc_c4_export_cells"):
        getattr(L, name).argtypes = [C.POINTER(OrcC4)]
    L.orc_c4_import.argtypes = [C.POINTER(OrcC4), vp]
    L.orc_c4_step.argtypes = [C.POINTER(OrcC4), i32]
    L.orc_c4_winner.argtypes = [C.POINTER(OrcC4)]; L.orc_c4_winner.restype = i32
    L.orc_c4_full.argtypes = [C.POINTER(OrcC4)]; L.orc_c4_full.restype = i32
    L.orc_c4_valid_moves.argtypes = [C.POINTER(OrcC4), vp]; L.orc_c4_valid_moves.restype = i32
    L.orc_c4_mirror.argtypes = [C.POINTER(OrcC4), i32]
    L.orc_c4_current_state.argtypes = [C.POINTER(OrcC4), vp]
    L.orc_mt_seed.argtypes = [C.POINTER(OrcMT), C.c_uint32]
    L.orc_mt_next.argtypes = [C.POINTER(OrcMT)]; L.orc_mt_next.restype = C.c_uint32
    L.orc_uniform_int.argtypes = [C.POINTER(OrcMT), i32, i32]; L.orc_uniform_int.restype = i32
    L.orc_canonical_float.argtypes = [C.POINTER(OrcMT)]; L.orc_canonical_float.restype = C.c_float
    L.orc_gamma_fill.argtypes = [C.POINTER(OrcMT), C.c_float, vp, i32]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class _Batched:
    """Same methods / dtypes / shapes as mcts_cpp.BatchedMCTS_<Game>."""
    PFX = "orc"
    action_size = A
    board_size = CELLS
    board_shape = (6, 7)

    def _f(self, name):
        return getattr(self._L, self.PFX + "_" + name)

    def __init__(self, n_envs):
        self._L = lib()
        self._h = self._f("create")(int(n_envs))
        self.n = int(n_envs)

    def __del__(self):
        if getattr(self, "_h", None):
            self._f("destroy")(self._h)
            self._h = None

    @property
    def config(self):
        return self._f("config_ptr")(self._h).contents

    def set_seed(self, seed):
        self._f("set_seed")(self._h, int(seed))

    def reset_env(self, i):
        self._f("reset_env")(self._h, int(i))

    def get_num_envs(self):
        return self.n

    def prune_roots(self, actions):
        a = _c(actions, np.int32)
        if a.ndim != 1 or a.size != self.n:
            raise RuntimeError("prune_roots: actions size must match n_envs")
        self._f("prune_roots")(self._h, _p(a))

    def search_batch(self, boards, turns):
        b = _c(boards, np.int8); t = _c(turns, np.int32)
        n = self.n
        if b.shape[0] != n or t.size != n:
            raise RuntimeError("search_batch: batch size must match n_envs")
        ob = np.empty((n,) + tuple(self.board_shape), np.int8)
        d, p1, p2 = (np.empty(n, np.float32) for _ in range(3))
        it = np.empty(n, np.uint8); ot = np.empty(n, np.int32); vm = np.empty((n, self.action_size), np.uint8)
        self._f("search_batch")(self._h, _p(b), _p(t), _p(ob), _p(d), _p(p1), _p(p2),
                                 _p(it), _p(ot), _p(vm))
        return ob, d, p1, p2, it, ot, vm

    def backprop_batch(self, policy_logits, d_vals, p1w_vals, p2w_vals, moves_left, is_term):
        pol = _c(policy_logits, np.float32); d = _c(d_vals, np.float32)
        p1 = _c(p1w_vals, np.float32); p2 = _c(p2w_vals, np.float32)
        ml = _c(moves_left, np.float32); it = _c(is_term, np.uint8)
        n = self.n
        if pol.shape[0] != n or any(x.size != n for x in (d, p1, p2, ml, it)):
            raise RuntimeError("backprop_batch: sizes must match n_envs")
        self._f("backprop_batch")(self._h, _p(pol), _p(d), _p(p1), _p(p2), _p(ml), _p(it))

    def remove_all_vl(self, K):
        self._f("remove_all_vl")(self._h, int(K))

    def search_batch_vl(self, K, input_boards, turns):
        b = _c(input_boards, np.int8); t = _c(turns, np.int32)
        n = self.n
        if b.shape[0] != n or t.size != n:
            raise RuntimeError("search_batch_vl: input batch != n_envs")
        if K < 1:
            raise RuntimeError("search_batch_vl: K must be >= 1")
        tot = n * K
        ob = np.empty((tot,) + tuple(self.board_shape), np.int8)
        d, p1, p2 = (np.empty(tot, np.float32) for _ in range(3))
        it = np.empty(tot, np.uint8); ot = np.empty(tot, np.int32)
        sy = np.empty(tot, np.int32); vm = np.empty((tot, self.action_size), np.uint8)
        self._f("search_batch_vl")(self._h, int(K), _p(b), _p(t), _p(ob), _p(d), _p(p1), _p(p2),
                                    _p(it), _p(ot), _p(sy), _p(vm))
        return ob, d, p1, p2, it, ot, sy, vm

    def backprop_batch_vl(self, K, policy_logits, d_vals, p1w_vals, p2w_vals, moves_left,
                          is_term, sym_ids):
        pol = _c(policy_logits, np.float32); d = _c(d_vals, np.float32)
        p1 = _c(p1w_vals, np.float32); p2 = _c(p2w_vals, np.float32)
        ml = _c(moves_left, np.float32); it = _c(is_term, np.uint8); sy = _c(sym_ids, np.int32)
        tot = self.n * K
        if pol.shape[0] != tot or any(x.size != tot for x in (d, p1, p2, ml, it, sy)):
            raise RuntimeError("backprop_batch_vl: sizes must be N*K")
        self._f("backprop_batch_vl")(self._h, int(K), _p(pol), _p(d), _p(p1), _p(p2), _p(ml),
                                      _p(it), _p(sy))

    def search_rollout(self, boards, turns, n_playout):
        b = _c(boards, np.int8); t = _c(turns, np.int32)
        self._f("search_rollout")(self._h, _p(b), _p(t), int(n_playout))

    def get_all_counts(self):
        out = np.empty(self.n * self.action_size, np.int32)
        self._f("get_all_counts")(self._h, _p(out))
        return out.tolist()

    def get_all_root_stats(self):
        out = np.empty((self.n, 6 + 8 * self.action_size), np.float32)
        self._f("get_all_root_stats")(self._h, _p(out))
        return out

    # --- oracle-only extras ---
    def pending_sym_ids(self):
        """Symmetry ids drawn by the last search_batch (pending_sym_ids_, BatchedMCTS.h:45)."""
        out = np.empty(self.n, np.int32)
        self._f("pending_sym")(self._h, _p(out))
        return out

    def stats(self):
        s = OrcStats()
        self._f("stats_get")(self._h, C.byref(s))
        return {n: getattr(s, n) for n, _ in OrcStats._fields_}

    def stats_reset(self):
        self._f("stats_reset")(self._h)

    def tree_size(self, env):
        a, b = C.c_int32(), C.c_int32()
        self._f("tree_size")(self._h, int(env), C.byref(a), C.byref(b))
        return a.value, b.value




class BatchedMCTS_Connect4(_Batched):
    PFX = "orc"
    action_size = 7
    board_size = 42
    board_shape = (6, 7)


class BatchedMCTS_Othello(_Batched):
    PFX = "oro"
    action_size = 65
    board_size = 64
    board_shape = (8, 8)


class Connect4Env:
    """Mirror of env_cpp.connect4.Env (env_common.h:133-249, env_connect4.h:29-65)."""
    NUM_SYMMETRIES = 2

    def __init__(self, board=None):
        self._L = lib()
        self.s = OrcC4()
        self._L.orc_c4_reset(C.byref(self.s))
        if board is not None:
            self.board = board

    def reset(self):
        self._L.orc_c4_reset(C.byref(self.s))

    def copy(self):
        e = Connect4Env()
        C.memmove(C.byref(e.s), C.byref(self.s), C.sizeof(OrcC4))
        return e

    def step(self, a):
        self._L.orc_c4_step(C.byref(self.s), int(a))

    def winPlayer(self):
        return self._L.orc_c4_winner(C.byref(self.s))

    check_winner = winPlayer

    def check_full(self):
        return bool(self._L.orc_c4_full(C.byref(self.s)))

    def done(self):
        return self.winPlayer() != 0 or self.check_full()

    @property
    def turn(self):
        return self.s.turn

    @turn.setter
    def turn(self, t):
        self.s.turn = int(t)

    @property
    def board(self):
        self._L.orc_c4_export_cells(C.byref(self.s))
        return np.array(self.s.cells, dtype=np.int8).reshape(6, 7).astype(np.float32)

    @board.setter
    def board(self, arr):
        a = _c(np.asarray(arr, dtype=np.float32).astype(np.int8), np.int8)
        if a.shape != (6, 7):
            raise RuntimeError("board shape must be (6, 7)")
        self._L.orc_c4_import(C.byref(self.s), _p(a))
        self.s.turn = 1 if self.s.n_pieces % 2 == 0 else -1

    def valid_move(self):
        m = np.empty(7, np.int32)
        n = self._L.orc_c4_valid_moves(C.byref(self.s), _p(m))
        return m[:n].tolist()

    def valid_mask(self):
        v = set(self.valid_move())
        return [a in v for a in range(A)]

    def current_state(self):
        out = np.empty((1, 3, 6, 7), np.float32)
        self._L.orc_c4_current_state(C.byref(self.s), _p(out))
        return out

    def apply_symmetry(self, sym_id, inplace=False):
        e = self if inplace else self.copy()
        e._L.orc_c4_mirror(C.byref(e.s), int(sym_id))
        return e

    @property
    def bitboards(self):
        return int(self.s.bb[0]), int(self.s.bb[1])


class OthelloEnv:
    """Mirror of env_cpp.othello.Env (env_common.h:133-249, env_othello.h:29-74)."""
    NUM_SYMMETRIES = 8

    def __init__(self, board=None):
        self._L = lib()
        self.s = OroOT()
        self._L.oro_ot_reset(C.byref(self.s))
        if board is not None:
            self.board = board

    def reset(self):
        self._L.oro_ot_reset(C.byref(self.s))

    def copy(self):
        e = OthelloEnv()
        C.memmove(C.byref(e.s), C.byref(self.s), C.sizeof(OroOT))
        return e

    def step(self, a):
        self._L.oro_ot_step(C.byref(self.s), int(a))

    def winPlayer(self):
        return self._L.oro_ot_winner(C.byref(self.s))

    check_winner = winPlayer

    def check_full(self):
        return bool(self._L.oro_ot_full(C.byref(self.s)))

    def done(self):
        return self.check_full()

    @property
    def turn(self):
        return self.s.turn

    @turn.setter
    def turn(self, t):
        self.s.turn = int(t)

    @property
    def board(self):
        self._L.oro_ot_export_cells(C.byref(self.s))
        return np.array(self.s.cells, dtype=np.int8).reshape(8, 8).astype(np.float32)

    @board.setter
    def board(self, arr):
        a = _c(np.asarray(arr, dtype=np.float32).astype(np.int8), np.int8)
        if a.shape != (8, 8):
            raise RuntimeError("board shape must be (8, 8)")
        self._L.oro_ot_import(C.byref(self.s), _p(a))
        self.s.turn = 1 if self.s.n_pieces % 2 == 0 else -1

    def valid_move(self):
        m = np.empty(65, np.int32)
        n = self._L.oro_ot_valid_moves(C.byref(self.s), _p(m))
        return m[:n].tolist()

    def valid_mask(self):
        v = set(self.valid_move())
        return [a in v for a in range(65)]

    def current_state(self):
        out = np.empty((1, 3, 8, 8), np.float32)
        self._L.oro_ot_current_state(C.byref(self.s), _p(out))
        return out

    def apply_symmetry(self, sym_id, inplace=False):
        e = self if inplace else self.copy()
        e._L.oro_ot_apply_sym(C.byref(e.s), int(sym_id))
        return e

    @property
    def bitboards(self):
        return int(self.s.bb[0]), int(self.s.bb[1])
