"""Search wrapper with the interface of the reference's src/MCTS_cpp.py (class BatchedMCTS,
MCTS_cpp.py:33-492), running on the MI355X engine behind `mcts_cpp`.

Two evaluation paths behind the same `batch_playout` call:

* host path - `pv_func.predict(states, action_mask)` on numpy arrays, exactly the
  reference's contract (MCTS_cpp.py:67-68, Connect4/Network.py:267-288).  Used for arbitrary
  evaluators, for the transposition cache, and for bit-exact parity runs; trees, selection,
  expansion and backup are HIP kernels, only the evaluator runs where the caller put it.
* fused path - when `pv_func` is a torch module living on the GPU the whole iteration
  (select -> feature gather -> network -> expand/backup) stays in HBM with no host
  synchronisation (see fused.py).  Chosen automatically; `fused=False` forces the host path.

There is no CPU search fallback: constructing BatchedMCTS without a usable GPU raises.
"""
import time
from collections import OrderedDict

import numpy as np

try:
    # PyTorch-ROCm wheels bundle their own libamdhip64; importing torch FIRST makes the engine
    # library bind to that same HIP runtime (one runtime per process: shared streams and
    # ordering with the network's kernels).  Without torch the system ROCm runtime is used.
    import torch  # noqa: F401
except ImportError:      # the search engine itself does not need torch
    pass

from src import mcts_cpp

# game name -> native class (MCTS_cpp.py:9-12)
_BACKENDS = {
    'Connect4': mcts_cpp.BatchedMCTS_Connect4,
}
if hasattr(mcts_cpp, 'BatchedMCTS_Othello'):
    _BACKENDS['Othello'] = mcts_cpp.BatchedMCTS_Othello


def _default_convert_board(board, turns):
    """Relative 3-plane features: own stones, opponent stones, side-to-move sign
    (MCTS_cpp.py:15-20)."""
    t = turns[:, None, None]
    own = (board == t).astype(np.float32)
    opp = (board == -t).astype(np.float32)
    side = np.ones_like(board, dtype=np.float32) * t
    return np.stack([own, opp, side], axis=1)


def _relative_wdl_to_absolute(wdl_rel, turns):
    """[draw, win, loss] of the side to move -> [draw, p1 win, p2 win] (MCTS_cpp.py:23-30)."""
    p1_to_move = turns == 1
    return (wdl_rel[:, 0],
            np.where(p1_to_move, wdl_rel[:, 1], wdl_rel[:, 2]),
            np.where(p1_to_move, wdl_rel[:, 2], wdl_rel[:, 1]))


class LRUCache:
    """Transposition table with the reference's exact LRU order (Cache.py:5-58): most recent
    at the FRONT of `_od`, eviction from the back."""

    def __init__(self, capacity=None):
        if capacity is None:
            self._cap = float('inf')
        elif int(capacity) == capacity and capacity >= 0:
            self._cap = capacity
        else:
            raise ValueError
        self._od = OrderedDict()

    @staticmethod
    def hash_ndarray(key):
        if isinstance(key, np.ndarray):
            return np.ascontiguousarray(key).tobytes()
        if isinstance(key, bytes):
            return key
        raise ValueError

    def __contains__(self, key):
        return self.hash_ndarray(key) in self._od

    def __len__(self):
        return len(self._od)

    def get(self, key):
        k = self.hash_ndarray(key)
        self._od.move_to_end(k, last=False)
        return self._od[k]['value']

    def put(self, key, value):
        k = self.hash_ndarray(key)
        if k in self._od:
            self._od[k]['value'] = value
            self._od.move_to_end(k, last=False)
            return
        self._od[k] = {'state': key, 'value': value}
        self._od.move_to_end(k, last=False)
        if len(self._od) > self._cap:
            self._od.popitem(last=True)

    def refresh(self, pv_func):
        if not self._od:
            return
        keys = list(self._od.keys())
        states = np.concatenate([self._od[k]['state'] for k in keys], axis=0)
        probs, values, moves_left = pv_func(states)
        for i, k in enumerate(keys):
            self.put(k, (probs[i].reshape(1, -1), values[i].reshape(1, -1), moves_left[i].reshape(1, -1)))


def _leaf_key(board, turn):
    """board bytes + one signed turn byte (MCTS_cpp.py:150)."""
    return board.tobytes() + int(turn).to_bytes(1, 'little', signed=True)


class BatchedMCTS:
    def __init__(self, batch_size, c_init, c_base, alpha, n_playout,
                 game_name='Connect4', board_converter=None, cache_size=0, noise_epsilon=0.25,
                 fpu_reduction=0.4, use_symmetry=True, mlh_slope=0.0, mlh_cap=0.2, value_decay=1.0,
                 score_utility_factor=0.0, score_scale=8.0):
        backend_cls = _BACKENDS[game_name]
        self.mcts = backend_cls(batch_size)

        cfg = self.mcts.config                      # live view of the engine's config
        cfg.c_init = c_init
        cfg.c_base = c_base
        cfg.dirichlet_alpha = alpha
        cfg.noise_epsilon = noise_epsilon
        cfg.fpu_reduction = fpu_reduction
        cfg.use_symmetry = use_symmetry
        cfg.mlh_slope = mlh_slope
        cfg.mlh_cap = mlh_cap
        cfg.score_utility_factor = score_utility_factor
        cfg.score_scale = score_scale
        cfg.value_decay = value_decay

        self.n_playout = n_playout
        self.batch_size = batch_size
        self.action_size = backend_cls.action_size
        self.board_shape = backend_cls.board_shape
        self._custom_converter = board_converter is not None
        self._convert_board = board_converter or _default_convert_board
        self.cache = LRUCache(cache_size) if cache_size > 0 else None
        self._cache_size = int(cache_size)
        self._game_name = game_name
        self._rollout_eval = None
        self._fused = None

    # ------------------------------------------------------------------ evaluation helpers
    def _predict_batch(self, pv_func, states, action_mask):
        return pv_func.predict(states, action_mask=action_mask)

    def _evaluate(self, pv_func, leaf_boards, leaf_turns, valid_masks, is_term, term_wdl, use_cache):
        """Fill policy / absolute WDL / moves-left for one batch of leaves: terminal leaves keep
        the engine's result with zero policy and zero moves-left, the others go to the
        transposition table and/or the evaluator (MCTS_cpp.py:117-189, 275-339)."""
        n = leaf_boards.shape[0]
        term_d, term_p1w, term_p2w = term_wdl
        d_vals, p1w_vals, p2w_vals = term_d.copy(), term_p1w.copy(), term_p2w.copy()
        moves_left = np.zeros(n, dtype=np.float32)
        probs = np.zeros((n, self.action_size), dtype=np.float32)
        live = ~is_term.astype(bool)
        if not live.any():
            return probs, d_vals, p1w_vals, p2w_vals, moves_left

        if not use_cache:
            turns = leaf_turns[live]
            conv = self._convert_board(leaf_boards[live], turns)
            p, wdl, ml = self._predict_batch(pv_func, conv, valid_masks[live].astype(bool, copy=False))
            probs[live] = p
            d_vals[live], p1w_vals[live], p2w_vals[live] = _relative_wdl_to_absolute(wdl, turns)
            moves_left[live] = ml.flatten()
            return probs, d_vals, p1w_vals, p2w_vals, moves_left

        def store(i, p, wdl, ml):
            probs[i] = p
            d_vals[i] = wdl[0]
            if leaf_turns[i] == 1:
                p1w_vals[i], p2w_vals[i] = wdl[1], wdl[2]
            else:
                p1w_vals[i], p2w_vals[i] = wdl[2], wdl[1]
            moves_left[i] = ml

        misses = []
        for i in np.where(live)[0]:
            key = _leaf_key(leaf_boards[i], leaf_turns[i])
            if key in self.cache:
                store(i, *self.cache.get(key))
            else:
                misses.append(i)
        if misses:
            m_turns = leaf_turns[misses]
            conv = self._convert_board(leaf_boards[misses], m_turns)
            m_masks = valid_masks[misses].astype(bool, copy=False)
            m_probs, m_wdl, m_ml = self._predict_batch(pv_func, conv, m_masks)
            m_ml = m_ml.flatten()
            for j, i in enumerate(misses):
                store(i, m_probs[j], m_wdl[j], m_ml[j])
                key = _leaf_key(leaf_boards[i], leaf_turns[i])
                self.cache.put(key, (m_probs[j].copy(), m_wdl[j].copy(), m_ml[j].item()))
                self.cache._od[key]['state'] = conv[j:j + 1]
                self.cache._od[key]['valid_mask'] = m_masks[j:j + 1].copy()
        return probs, d_vals, p1w_vals, p2w_vals, moves_left

    def _plain_iteration(self, pv_func, boards, turns, use_cache):
        lb, td, t1, t2, is_term, lt, vm = self.mcts.search_batch(boards, turns)
        probs, d, p1, p2, ml = self._evaluate(pv_func, lb, lt, vm, is_term, (td, t1, t2), use_cache)
        c = np.ascontiguousarray
        self.mcts.backprop_batch(c(probs, dtype=np.float32), c(d, dtype=np.float32),
                                 c(p1, dtype=np.float32), c(p2, dtype=np.float32),
                                 c(ml, dtype=np.float32), is_term)

    def _vl_iteration(self, pv_func, boards, turns, k):
        lb, td, t1, t2, is_term, lt, sym_ids, vm = self.mcts.search_batch_vl(k, boards, turns)
        try:
            probs, d, p1, p2, ml = self._evaluate(pv_func, lb, lt, vm, is_term, (td, t1, t2),
                                                  self.cache is not None)
            c = np.ascontiguousarray
            self.mcts.backprop_batch_vl(k, c(probs, dtype=np.float32), c(d, dtype=np.float32),
                                        c(p1, dtype=np.float32), c(p2, dtype=np.float32),
                                        c(ml, dtype=np.float32), is_term, sym_ids)
        except BaseException:
            # evaluator failed / interrupted: leave no in-flight visits behind (MCTS_cpp.py:351-355)
            self.mcts.remove_all_vl(k)
            raise

    def _should_early_exit(self, step, remaining_steps):
        """True when in every tree the runner-up can no longer catch the most visited action
        within the remaining budget (MCTS_cpp.py:70-87)."""
        if step < 8:
            return False
        counts = np.array(self.mcts.get_all_counts()).reshape(self.batch_size, self.action_size)
        top2 = np.partition(counts, -2, axis=1)[:, -2:]
        return bool(np.all(top2.max(axis=1) - top2.min(axis=1) > remaining_steps))

    # ------------------------------------------------------------------ search
    def _fused_runner(self, pv_func, fused):
        """The device-resident loop applies when the evaluator is a torch module on the GPU and
        nothing forces the host contract (custom features; the LRU transposition cache, unless
        the caller passes fused=True, which moves the cache into a device table of at least
        `cache_size` entries - include/az_mcts.h)."""
        if fused is False or self._custom_converter:
            return None
        if self.cache is not None and fused is not True:
            return None
        try:
            from src import fused as _fused
        except ImportError:
            if fused:
                raise
            return None
        if not _fused.is_device_module(pv_func):
            if fused:
                raise RuntimeError("fused=True needs a torch.nn.Module evaluator on a GPU device")
            return None
        if self._fused is None or self._fused.net is not pv_func:
            self._fused = _fused.FusedSearch(self, pv_func)
            if self.cache is not None:
                want = max(self._cache_size, 4)
                self._fused.enable_table(min(max((want - 1).bit_length(), 10), 26))
        return self._fused

    def batch_playout(self, pv_func, current_boards, turns, n_playout=None,
                      vl_batch=1, time_budget=None, fused=None):
        """n_playout simulations on every tree (or until time_budget seconds are used).

        current_boards [batch, *board_shape] with X=1 / O=-1, turns [batch] in {1,-1};
        vl_batch > 1 uses virtual-loss batches after one warm-up simulation
        (MCTS_cpp.py:89-359)."""
        current_boards = np.asarray(current_boards).astype(np.int8)
        turns = np.asarray(turns).astype(np.int32)
        max_n = n_playout if n_playout is not None else self.n_playout
        use_time = time_budget is not None and time_budget > 0

        if hasattr(pv_func, 'score_scale'):          # MCTS_cpp.py:106-108
            pv_func.score_scale = self.mcts.config.score_scale

        runner = self._fused_runner(pv_func, fused)
        if runner is not None:
            if use_time:       # wall-clock check and top-2 early exit between chunks of whole iterations (fused.py search_timed)
                self.last_playouts = runner.playout_timed(current_boards, turns, max_n, vl_batch, time_budget)
            else:
                runner.playout(current_boards, turns, max_n, vl_batch)
                self.last_playouts = max_n
            return self

        t0 = time.perf_counter() if use_time else 0.0
        if vl_batch <= 1:
            for step in range(max_n):
                self._plain_iteration(pv_func, current_boards, turns, self.cache is not None)
                if use_time:
                    done = step + 1
                    elapsed = time.perf_counter() - t0
                    if elapsed >= time_budget:
                        break
                    if self._should_early_exit(done, (time_budget - elapsed) / (elapsed / done)):
                        break
            return self

        remaining = max_n
        total_sims = 0
        if remaining > 0:
            # warm-up: one plain simulation so that every root is expanded before K virtual-loss
            # descents share it; it never consults the transposition table (MCTS_cpp.py:217-248)
            self._plain_iteration(pv_func, current_boards, turns, False)
            remaining -= 1
            total_sims += 1
        while remaining > 0:
            if use_time:
                elapsed = time.perf_counter() - t0
                if elapsed >= time_budget:
                    break
                if total_sims > 0 and self._should_early_exit(
                        total_sims, (time_budget - elapsed) / (elapsed / total_sims)):
                    break
            k = min(vl_batch, remaining)
            remaining -= k
            self._vl_iteration(pv_func, current_boards, turns, k)
            total_sims += k
        return self

    run = batch_playout   # alias named in BASELINE.json's north_star ("MCTSBatch.run")

    def refresh_cache(self, pv_func):
        """Re-evaluate every cached position after a weight update (MCTS_cpp.py:361-377).  The
        device table of the fused path does the same in HBM (az_mcts_dev_tt_refresh: every resident
        key evaluated again with the module's current weights, keys and ages kept)."""
        if self._fused is not None and self._fused.table_log2:
            seen = self._fused._fast_version
            self._fused._sync_fast_net()            # new snapshot of the weights; refreshes if they changed
            if self._fused._fast_version == seen:
                self._fused.refresh_table()
        if self.cache is None or len(self.cache) == 0:
            return self
        if hasattr(pv_func, 'score_scale'):
            pv_func.score_scale = self.mcts.config.score_scale
        od = self.cache._od
        keys = list(od.keys())
        states = np.concatenate([od[k]['state'] for k in keys], axis=0)
        masks = None
        if all('valid_mask' in od[k] for k in keys):
            masks = np.concatenate([od[k]['valid_mask'] for k in keys], axis=0)
        probs, wdl, ml = self._predict_batch(pv_func, states, masks)
        ml = ml.flatten()
        for j, k in enumerate(keys):
            od[k]['value'] = (probs[j].copy(), wdl[j].copy(), ml[j].item())
        return self

    def _get_rollout_evaluator(self):
        if self._rollout_eval is None:
            self._rollout_eval = getattr(mcts_cpp, f'RolloutEvaluator_{self._game_name}')()
        return self._rollout_eval

    def rollout_playout(self, current_boards, turns):
        """Pure MCTS with random playouts, whole loop inside the engine (MCTS_cpp.py:386-392)."""
        current_boards = np.asarray(current_boards).astype(np.int8)
        turns = np.asarray(turns).astype(np.int32)
        self.mcts.search(self._get_rollout_evaluator(), current_boards, turns, self.n_playout)
        return self

    # ------------------------------------------------------------------ config setters (MCTS_cpp.py:394-427)
    def set_noise_epsilon(self, eps):
        self.mcts.config.noise_epsilon = eps

    def set_mlh_params(self, slope, cap):
        cfg = self.mcts.config
        cfg.mlh_slope = slope
        cfg.mlh_cap = cap

    def set_score_utility_params(self, factor, scale):
        cfg = self.mcts.config
        old_scale = cfg.score_scale
        cfg.score_utility_factor = factor
        cfg.score_scale = scale
        if scale != old_scale and self.cache is not None and len(self.cache) > 0:
            self.cache._od.clear()

    def set_c_init(self, val):
        self.mcts.config.c_init = val

    def set_c_base(self, val):
        self.mcts.config.c_base = val

    def set_alpha(self, val):
        self.mcts.config.dirichlet_alpha = val

    def set_fpu_reduction(self, val):
        self.mcts.config.fpu_reduction = val

    def set_use_symmetry(self, val):
        self.mcts.config.use_symmetry = val

    def set_value_decay(self, val):
        self.mcts.config.value_decay = val

    # ------------------------------------------------------------------ tree management / queries
    def reset_env(self, index):
        self.mcts.reset_env(index)
        return self

    def seed(self, seed):
        self.mcts.set_seed(seed)

    def prune_roots(self, actions):
        self.mcts.prune_roots(np.ascontiguousarray(actions, dtype=np.int32))
        return self

    def get_visits_count(self):
        return np.array(self.mcts.get_all_counts()).reshape(self.batch_size, self.action_size)

    def get_mcts_probs(self):
        counts = self.get_visits_count()
        return counts / counts.sum(axis=1, keepdims=True)

    def get_root_stats(self):
        """Root statistics of every tree as a dict of arrays (MCTS_cpp.py:449-492): root_N,
        root_Q, root_M, root_D, root_P1W, root_P2W of shape (batch,), and per action N, Q,
        prior, noise, M, D, P1W, P2W of shape (batch, action_size); WDL in the absolute view."""
        raw = self.mcts.get_all_root_stats()
        head = raw[:, :6]
        per_action = raw[:, 6:].reshape(self.batch_size, self.action_size, 8)
        out = {name: head[:, i] for i, name in
               enumerate(('root_N', 'root_Q', 'root_M', 'root_D', 'root_P1W', 'root_P2W'))}
        out.update({name: per_action[:, :, i] for i, name in
                    enumerate(('N', 'Q', 'prior', 'noise', 'M', 'D', 'P1W', 'P2W'))})
        return out
