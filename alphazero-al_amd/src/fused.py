"""Device-resident search loop: select -> leaf gather -> network -> expand/backup with no host
round trip.

The reference crosses Python<->C++ twice and host<->device twice per MCTS iteration
(src/MCTS_cpp.py:250-357, Connect4/Network.py:267-288).  Here one iteration is
    az_mcts_dev_select     HIP: K descents per tree + gather of the leaves into the
                           evaluator's (n,3,6,7) input and (n,7) action mask, all in HBM
    net(...)               PyTorch-ROCm on the same stream (bf16 autocast as the reference)
    az_mcts_dev_backprop   HIP: relative->absolute WDL, expansion, backup
enqueued back to back on the current stream, and - once shapes are warm - replayed from a
hipGraph (torch.cuda.CUDAGraph), so the host issues one graph launch per iteration.

Randomness (symmetry ids, Dirichlet noise) comes from the engine's counter-based device
generator on this path; the reference's own stream (host mt19937) is what the host path draws.
With use_symmetry=False and dirichlet_alpha<=0 nothing random is consumed and this path is
bit-identical to the host path (tests/test_fused_gpu.py); with randomness on, `FusedSearch.replay`
plays recorded draws back so that the loop can be compared with the oracle bit for bit
(tests/test_replay_gpu.py).
"""
import ctypes as C
import gc
import weakref
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(os.path.join(os.path.dirname(_HERE), "lib", "libaz_mcts.so"))
        vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
        L.az_last_error.restype = C.c_char_p
        L.az_mcts_dev_prepare.argtypes = [vp, i32, i64]
        L.az_mcts_dev_prepare_stream.argtypes = [vp, i32, i64, vp]
        L.az_mcts_dev_check.argtypes = [vp, vp]
        L.az_mcts_dev_replay.argtypes = [vp, vp, i64, i64, vp]
        L.az_nn_model_create_hash.argtypes = [i32, C.POINTER(vp)]
        L.az_nn_model_destroy.argtypes = [vp]
        L.az_mcts_dev_set_roots.argtypes = [vp, vp, vp, vp, vp]
        L.az_mcts_dev_import_roots.argtypes = [vp, vp, vp, vp]
        L.az_mcts_dev_select.argtypes = [vp, i32, i32, vp, vp, vp]
        L.az_mcts_dev_backprop.argtypes = [vp, i32, i32, vp, vp, vp, vp]
        L.az_mcts_dev_leaves.argtypes = [vp, i32, vp, vp, vp, vp, vp]
        L.az_mcts_dev_leaf_syms.argtypes = [vp, i32, vp, vp]
        L.az_mcts_dev_counts.argtypes = [vp, vp, vp]
        L.az_mcts_dev_root_stats.argtypes = [vp, vp, vp]
        L.az_mcts_dev_prune_roots.argtypes = [vp, vp, vp]
        L.az_mcts_dev_reset_masked.argtypes = [vp, vp, vp]
        L.az_mcts_reserve.argtypes = [vp, i64]
        L.az_mcts_capacity.argtypes = [vp]; L.az_mcts_capacity.restype = i64
        L.az_mcts_epoch.argtypes = [vp]; L.az_mcts_epoch.restype = i64
        L.az_mcts_max_used.argtypes = [vp, C.POINTER(i64)]
        L.az_mcts_counters.argtypes = [vp, C.POINTER(i64 * 8)]
        L.az_mcts_counters_reset.argtypes = [vp]
        L.az_mcts_profile.argtypes = [vp, i32]
        L.az_mcts_profile_read.argtypes = [vp, C.POINTER(C.c_double * 2), C.POINTER(i64 * 2)]
        L.az_c4_dev_step.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, vp]
        L.az_game_dev_step.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp]
        L.az_game_dev_valid_mask.argtypes = [i32, vp, vp, vp, vp, vp, i64, vp]
        L.az_mcts_dev_live_leaves.argtypes = [vp, i32, vp, vp, vp]
        L.az_mcts_dev_search.argtypes = [vp, vp, i32, i32, i32, vp]
        L.az_mcts_dev_search_more.argtypes = [vp, vp, i32, i32, i32, vp]
        L.az_mcts_dev_tt_create.argtypes = [vp, i32]
        L.az_mcts_dev_tt_clear.argtypes = [vp, vp]
        L.az_mcts_dev_tt_lookup.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
        L.az_mcts_dev_tt_insert.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
        L.az_mcts_dev_tt_stats.argtypes = [vp, C.POINTER(i64 * 4)]
        L.az_mcts_dev_tt_refresh.argtypes = [vp, vp, vp]
        _LIB = L
    return _LIB


def hip_runtimes_loaded():
    """Paths of the libamdhip64 images mapped into this process."""
    try:
        with open("/proc/self/maps") as f:
            return sorted({ln.split()[-1] for ln in f if "libamdhip64" in ln})
    except OSError:
        return []


def require_single_hip_runtime():
    """The fused loop shares streams between PyTorch and the engine, which is only meaningful
    inside ONE HIP runtime.  Two get loaded when the engine library is imported before torch
    (torch then brings its bundled copy): fail loudly instead of racing."""
    libs = hip_runtimes_loaded()
    if len(libs) > 1:
        raise RuntimeError("two HIP runtimes are loaded (%s): import torch before src.mcts_cpp / "
                           "src.MCTS_cpp so that the engine binds to torch's runtime" % ", ".join(libs))


def check(rc):
    if rc != 0:
        raise RuntimeError(lib().az_last_error().decode())


def is_device_module(pv):
    if not isinstance(pv, torch.nn.Module):
        return False
    p = next(pv.parameters(), None)
    if p is None:
        p = next(pv.buffers(), None)
    if p is None:
        return bool(getattr(pv, "is_device_evaluator", False))
    return p.is_cuda


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def counters(handle):
    arr = (C.c_int64 * 8)()
    check(lib().az_mcts_counters(handle, C.byref(arr)))
    names = ("sims", "levels", "expansions", "terminal", "dup_leaves", "backup_nodes",
             "select_launches", "backprop_launches")
    return dict(zip(names, list(arr)))


class FusedSearch:
    """One instance per (BatchedMCTS wrapper, network) pair."""

    def __init__(self, wrapper, net):
        require_single_hip_runtime()
        # no reference cycle with the wrapper (which owns this object): an engine must be freed by
        # reference counting, never by a cyclic-GC pass that happens to run inside a graph capture
        self._w = weakref.ref(wrapper)
        self.mcts = wrapper.mcts
        self.board_shape = tuple(wrapper.board_shape)
        self.game_name = wrapper._game_name
        self.net = net
        self.h = C.c_void_p(wrapper.mcts.handle)
        self.B = wrapper.batch_size
        self.A = wrapper.action_size
        p = next(net.parameters(), None)
        self.device = p.device if p is not None else torch.device("cuda", torch.cuda.current_device())
        # hipGraph replay pays when launches, not kernels, bound an iteration: small leaf batches
        g = os.environ.get("AZ_FUSED_GRAPH", "auto")
        self.use_graph = (self.B <= 512) if g == "auto" else g != "0"
        self.autocast = os.environ.get("AZ_FUSED_AUTOCAST", "1") != "0"
        self._bufs = {}      # K -> (features, mask)
        self._graphs = {}    # (K, vl, cfg key, epoch) -> CUDAGraph
        self._eager_runs = {}
        self.aux_scale = float(getattr(net, "aux_target_offset", 1.0))
        # inference twin of the reference architecture (fast_net.py); AZ_FUSED_FASTNET=0 calls
        # the caller's module under autocast exactly as the reference's predict() does
        self.fast = None
        self._fast_version = None
        self.use_fast = os.environ.get("AZ_FUSED_FASTNET", "1") != "0"
        # device transposition table (enable_table): off unless asked for
        self.table_log2 = 0
        self.compact_eval = os.environ.get("AZ_FUSED_COMPACT", "1") != "0"
        self.table_verify = False
        self._tt_bufs = {}
        self.tt_mismatch = None
        self._hash_model = None

    def _sync_fast_net(self):
        if not self.use_fast or hasattr(self.net, "predict_device"):
            return
        from src.fast_net import FastConnect4Net
        from src.fast_othello import FastOthelloNet
        if self.game_name == "Othello" and FastOthelloNet.recognises(self.net):
            make = lambda: FastOthelloNet(self.net)                                   # noqa: E731
        elif self.game_name != "Othello" and FastConnect4Net.recognises(self.net):
            make = lambda: FastConnect4Net.from_module(self.net, device=self.device)  # noqa: E731
        else:
            return
        version = (tuple(p._version for p in self.net.parameters()) + tuple(p.data_ptr() for p in self.net.parameters()) +
                   tuple(b._version for b in self.net.buffers()))
        if self.fast is None or version != self._fast_version:
            self.fast = make()
            self._fast_version = version
            self._graphs.clear()        # captured graphs hold the old weight buffers
            self._eager_runs.clear()
            if self.table_log2:         # cached outputs belong to the old weights (MCTS_cpp.py:361-377)
                self.refresh_table()

    def _native_model(self):
        """az_nn_model* when the search can run as one native call: HIP inference twin, compact
        evaluation on, no verify pass, no graph replay asked for (AZ_FUSED_NATIVE=0 keeps the
        Python loop over the same entry points)."""
        if self.use_graph or self.table_verify or not self.compact_eval or os.environ.get("AZ_FUSED_NATIVE", "1") == "0":
            return None
        game = getattr(self.net, "native_hash_game", None)
        if game is not None:
            # the integer-hash evaluator has a native twin for either game (az_nn_model_create_hash)
            if self._hash_model is None:
                h = C.c_void_p()
                if lib().az_nn_model_create_hash(int(game), C.byref(h)) != 0:
                    raise RuntimeError("az_nn_model_create_hash failed")
                self._hash_model = h
            return self._hash_model
        if self.fast is None or not hasattr(self.fast, "native_model"):
            return None
        return self.fast.native_model()

    def __del__(self):
        try:
            h = self.__dict__.get("_hash_model")
            if h is not None and _LIB is not None:
                self.__dict__["_hash_model"] = None
                _LIB.az_nn_model_destroy(h)
        except Exception:
            pass

    def replay(self, sym_ids=None, root_noise=None):
        """Recorded draws instead of the device generator (az_mcts_dev_replay, parity tests):
        sym_ids int32 tensor (n_select_calls, stride >= B*K) on the device, root_noise float32
        (B, A) by edge index; both None ends the replay.  The tensors are kept alive here."""
        self._replay_keep = (sym_ids, root_noise)
        sp = sym_ids.data_ptr() if sym_ids is not None else None
        stride, calls = (sym_ids.shape[1], sym_ids.shape[0]) if sym_ids is not None else (0, 0)
        check(lib().az_mcts_dev_replay(self.h, sp, stride, calls, root_noise.data_ptr() if root_noise is not None else None))

    # ------------------------------------------------------------------ transposition table
    def enable_table(self, log2_entries=20, verify=False):
        """Keep evaluator outputs of positions already seen in a device hash table (include/az_mcts.h;
        the reference's `cache_size`, src/Cache.py).  Needs an evaluator that can work on a compact
        list of rows (the HIP inference twin); with it an iteration costs the leaves that MISS.
        `verify=True` also evaluates every leaf densely and counts rows whose table value differs
        from the fresh one (`tt_mismatch`, must stay 0: the evaluator is a pure function per row)."""
        self._sync_fast_net()
        native_hash = getattr(self.net, "native_hash_game", None) is not None and self._native_model() is not None
        if not native_hash and (self.fast is None or not getattr(self.fast, "supports_compact", False)):
            raise RuntimeError("the device transposition table needs the HIP inference twin of the Connect4 network "
                               "(a module with the reference CNN's parameters on a GPU)")
        check(lib().az_mcts_dev_tt_create(self.h, int(log2_entries)))
        if getattr(self.fast, "compact_needs_host", False):
            self.use_graph = False        # the miss list's length is read back on the host: nothing to capture
        self.table_log2 = int(log2_entries)
        self.table_verify = bool(verify)
        self.tt_mismatch = torch.zeros((), dtype=torch.int64, device=self.device)
        self._graphs.clear()
        self._eager_runs.clear()

    def _table_row_width(self):
        return self.A

    def refresh_table(self):
        """`refresh_cache` of the reference's wrapper (MCTS_cpp.py:361-377) for the device table: every
        resident key is evaluated again with the current weights; evaluators without a native model
        object get an empty table instead."""
        if not self.table_log2:
            return
        model = None
        if getattr(self.net, "native_hash_game", None) is not None:
            model = self._native_model()
        elif self.fast is not None and hasattr(self.fast, "native_model"):
            model = self.fast.native_model()
        if model is not None:
            check(lib().az_mcts_dev_tt_refresh(self.h, model, _stream()))
        else:                                        # an evaluator without a native model object
            check(lib().az_mcts_dev_tt_clear(self.h, _stream()))

    def table_stats(self):
        arr = (C.c_int64 * 4)()
        check(lib().az_mcts_dev_tt_stats(self.h, C.byref(arr)))
        d = dict(zip(("lookups", "hits", "inserts", "replaced"), list(arr)))
        d["hit_rate"] = d["hits"] / d["lookups"] if d["lookups"] else 0.0
        if self.tt_mismatch is not None:
            d["mismatches"] = int(self.tt_mismatch.item())
        return d

    def _table_buffers(self, K):
        if K not in self._tt_bufs:
            n = self.B * K
            z = dict(device=self.device)
            self._tt_bufs[K] = (torch.zeros((n, self.A), dtype=torch.float32, **z), torch.zeros((n, 3), dtype=torch.float32, **z),
                                torch.zeros((n,), dtype=torch.float32, **z), torch.zeros((n,), dtype=torch.int32, **z),
                                torch.zeros((1,), dtype=torch.int64, **z))
        return self._tt_bufs[K]

    def _iteration_with_table(self, K, vl):
        L = lib()
        feats, mask = self._buffers(K)
        probs, wdl, ml, rows, n_rows = self._table_buffers(K)
        s = _stream()
        check(L.az_mcts_dev_select(self.h, K, vl, feats.data_ptr(), mask.data_ptr(), s))
        check(L.az_mcts_dev_tt_lookup(self.h, K, probs.data_ptr(), wdl.data_ptr(), ml.data_ptr(), rows.data_ptr(),
                                      n_rows.data_ptr(), s))
        self.fast.predict_device(feats, mask, rows=rows, n_rows=n_rows, out=(probs, wdl, ml))
        check(L.az_mcts_dev_tt_insert(self.h, K, rows.data_ptr(), n_rows.data_ptr(), probs.data_ptr(), wdl.data_ptr(),
                                      ml.data_ptr(), s))
        keep = None
        if self.table_verify:
            p2, w2, m2 = self.fast.predict_device(feats, mask)
            live = mask.view(torch.uint8).any(dim=1)                 # terminal leaves show an all-zero mask
            bad = ((probs != p2).any(1) | (wdl != w2).any(1) | (ml != m2)) & live
            self.tt_mismatch += bad.sum()
            keep = (p2, w2, m2)
        check(L.az_mcts_dev_backprop(self.h, K, vl, probs.data_ptr(), wdl.data_ptr(), ml.data_ptr(), s))
        return keep

    # ------------------------------------------------------------------ pieces
    def _buffers(self, K):
        if K not in self._bufs:
            n = self.B * K
            self._bufs[K] = (torch.empty((n, 3) + self.board_shape, dtype=torch.float32, device=self.device),
                             torch.empty((n, self.A), dtype=torch.uint8, device=self.device))
        return self._bufs[K]

    def evaluate(self, features, mask_u8):
        """What `predict` computes (Connect4/Network.py:267-288), kept on the device."""
        mask = mask_u8.view(torch.bool) if mask_u8.dtype == torch.uint8 else mask_u8
        if hasattr(self.net, "predict_device"):
            probs, wdl, ml = self.net.predict_device(features, mask)
        elif self.fast is not None:
            probs, wdl, ml = self.fast.predict_device(features, mask_u8)
        else:
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.autocast):
                log_prob, value_lp, steps = self.net(features, action_mask=mask)
            probs = log_prob.float().exp()
            wdl = value_lp.exp().float()
            ml = (steps * self.aux_scale).float()
            if self.game_name == "Othello":
                # the Othello head predicts a disc difference; the search consumes its utility
                # (Othello/Network.py:247-249)
                scale = float(getattr(self.net, "score_scale", 8.0))
                ml = torch.atan(ml / scale) * (2.0 / 3.141592653589793)
        return probs.contiguous(), wdl.contiguous(), ml.reshape(-1).contiguous()

    def _iteration_compact(self, K, vl):
        """Evaluator on the non-terminal leaves only (the reference's wrapper does the same,
        MCTS_cpp.py:275-297): late in a game a sizeable share of the leaves is terminal."""
        L = lib()
        feats, mask = self._buffers(K)
        probs, wdl, ml, rows, n_rows = self._table_buffers(K)
        s = _stream()
        check(L.az_mcts_dev_select(self.h, K, vl, feats.data_ptr(), mask.data_ptr(), s))
        check(L.az_mcts_dev_live_leaves(self.h, K, rows.data_ptr(), n_rows.data_ptr(), s))
        self.fast.predict_device(feats, mask, rows=rows, n_rows=n_rows, out=(probs, wdl, ml))
        check(L.az_mcts_dev_backprop(self.h, K, vl, probs.data_ptr(), wdl.data_ptr(), ml.data_ptr(), s))
        return None

    def _iteration(self, K, vl):
        if self.table_log2:
            return self._iteration_with_table(K, vl)
        if (self.compact_eval and self.fast is not None and getattr(self.fast, "supports_compact", False)
                and not (self.use_graph and getattr(self.fast, "compact_needs_host", False))):
            return self._iteration_compact(K, vl)
        L = lib()
        feats, mask = self._buffers(K)
        s = _stream()
        check(L.az_mcts_dev_select(self.h, K, vl, feats.data_ptr(), mask.data_ptr(), s))
        probs, wdl, ml = self.evaluate(feats, mask)
        check(L.az_mcts_dev_backprop(self.h, K, vl, probs.data_ptr(), wdl.data_ptr(), ml.data_ptr(), s))
        return probs, wdl, ml      # keep alive until enqueued work is ordered behind them

    def _cfg_key(self):
        c = self.mcts.config
        return (c.c_init, c.c_base, c.dirichlet_alpha, c.noise_epsilon, c.fpu_reduction, c.mlh_slope,
                c.mlh_cap, c.value_decay, bool(c.use_symmetry), c.vl_count)

    def _run(self, K, vl):
        """One iteration: eager for the first two runs of a shape (lets MIOpen / hipBLASLt pick
        kernels and allocate workspaces), then captured once and replayed."""
        if not self.use_graph:
            self._keep = self._iteration(K, vl)
            return
        key = (K, vl, self._cfg_key(), lib().az_mcts_epoch(self.h))
        g = self._graphs.get(key)
        if g is not None:
            g.replay()
            return
        runs = self._eager_runs.get(key, 0)
        if runs < 2:
            self._eager_runs[key] = runs + 1
            self._keep = self._iteration(K, vl)
            return
        # drop graphs of stale epochs / configs for this shape
        for k in [k for k in self._graphs if k[0] == K and k[1] == vl]:
            del self._graphs[k]
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        # Nothing may free device memory or synchronise while the stream is capturing: a garbage
        # collection that finalises some other engine or graph in there aborts the process.
        gc_was_on = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                outs = self._iteration(K, vl)
        finally:
            if gc_was_on:
                gc.enable()
        self._graphs[key] = g
        self._graph_outs = getattr(self, "_graph_outs", {})
        self._graph_outs[key] = outs
        g.replay()           # capture does not execute: this is the real iteration

    # ------------------------------------------------------------------ public
    def upload_roots(self, boards, turns):
        b = torch.from_numpy(np.ascontiguousarray(boards, dtype=np.int8)).to(self.device, non_blocking=False)
        t = torch.from_numpy(np.ascontiguousarray(turns, dtype=np.int32)).to(self.device, non_blocking=False)
        check(lib().az_mcts_dev_import_roots(self.h, b.data_ptr(), t.data_ptr(), _stream()))
        self._roots_keep = (b, t)

    def search(self, n_playout, vl_batch):
        """The reference's iteration schedule (MCTS_cpp.py:110-113, 217-264) on roots that are
        already in HBM."""
        K = max(1, int(vl_batch))
        self._sync_fast_net()
        model = self._native_model()
        if model is not None:
            # the whole schedule, evaluator included, from native code (az_mcts_dev_search)
            check(lib().az_mcts_dev_search(self.h, model, int(n_playout), K, 1 if self.table_log2 else 0, _stream()))
            return
        check(lib().az_mcts_dev_prepare_stream(self.h, K, int(n_playout), _stream()))
        if K <= 1:
            for _ in range(n_playout):
                self._run(1, 0)
            return
        remaining = n_playout
        if remaining > 0:
            self._run(1, 0)          # warm-up simulation expands every root
            remaining -= 1
        while remaining > 0:
            k = min(K, remaining)
            remaining -= k
            self._run(k, 1)

    def search_timed(self, max_n, vl_batch, time_budget, early_exit=True, chunk_fraction=0.1):
        """The reference's search under a TIME budget (MCTS_cpp.py:194-209 plain, 252-264 virtual loss) on the device
        loop: before every chunk of whole iterations the wall clock is read and, from 8 simulations on, the top-2 test
        of `_should_early_exit` (MCTS_cpp.py:70-87) is made on the root visit counts - which costs ONE drain of the
        stream per chunk (the counts travel to pinned host memory behind the chunk's kernels).  A chunk is sized to
        ~`chunk_fraction` of the budget from the measured time per simulation, so the search overshoots the budget by
        at most about that share; `max_n` bounds the simulations as n_playout does.  Returns the simulations run.
        Every chunk is whole iterations (selection + backup): no in-flight visits are left behind."""
        import time
        K = max(1, int(vl_batch))
        t0 = time.perf_counter()
        self._sync_fast_net()
        model = self._native_model()
        L, s = lib(), _stream()
        if getattr(self, "_counts_dev", None) is None:
            self._counts_dev = torch.zeros((self.B, self.A), dtype=torch.int32, device=self.device)
            self._counts_host = torch.zeros((self.B, self.A), dtype=torch.int32).pin_memory()
        table = 1 if self.table_log2 else 0

        def run(n, first):
            if model is not None:
                fn = L.az_mcts_dev_search if first else L.az_mcts_dev_search_more
                check(fn(self.h, model, int(n), K, table, s))
                return
            check(L.az_mcts_dev_prepare_stream(self.h, K, int(n), s))
            left = n
            if K <= 1 or first:
                self._run(1, 0)
                left -= 1
            while left > 0:
                k = min(K, left) if K > 1 else 1
                left -= k
                self._run(k, 1 if K > 1 else 0)

        def counts():
            check(L.az_mcts_dev_counts(self.h, self._counts_dev.data_ptr(), s))
            self._counts_host.copy_(self._counts_dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            ev.synchronize()                                  # the one drain of this chunk
            return self._counts_host.numpy()

        total, remaining = 0, int(max_n)
        if remaining > 0:
            run(1, True)                                      # the warm-up simulation (plain path: the first one)
            total, remaining = 1, remaining - 1
        while remaining > 0:
            c = counts()
            elapsed = time.perf_counter() - t0
            if elapsed >= time_budget:
                break
            per_sim = elapsed / total
            if early_exit and total >= 8:
                top2 = np.partition(c, -2, axis=1)[:, -2:]
                if bool(np.all(top2.max(axis=1) - top2.min(axis=1) > (time_budget - elapsed) / per_sim)):
                    break
            # whole iterations worth about a tenth of the budget, never past the budget's end by more than one chunk
            want = min(chunk_fraction * time_budget, time_budget - elapsed) / per_sim
            n = int(min(remaining, max(K, (int(want) // K) * K)))
            run(n, False)
            total += n
            remaining -= n
        return total

    def playout_timed(self, boards, turns, max_n, vl_batch, time_budget):
        with torch.cuda.device(self.device):
            self.upload_roots(boards, turns)
            done = self.search_timed(max_n, vl_batch, time_budget)
            torch.cuda.current_stream().synchronize()           # host entry points use the NULL stream
        return done

    def playout(self, boards, turns, n_playout, vl_batch):
        with torch.cuda.device(self.device):
            self.upload_roots(boards, turns)
            self.search(n_playout, vl_batch)
            if torch.cuda.current_stream().cuda_stream != 0:
                torch.cuda.current_stream().synchronize()   # host entry points use the NULL stream
