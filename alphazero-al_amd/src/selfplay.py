"""Device-side self-play driver: thousands of games (Connect4 or Othello) and their search trees
advance in lockstep without leaving HBM.

The reference's driver (src/game.py:65-164 with player.py:333-375) is O(batch) Python per ply
- it rebuilds the numpy boards from per-game Env objects, samples each action in a Python
loop and keeps finished games in the batch until the slowest one ends.  Here a ply is
    roots (bitboards in HBM) -> FusedSearch.search (n_playout simulations per tree)
    -> root visit counts (HIP) -> temperature sampling (torch, on device)
    -> prune_roots with fresh device Dirichlet noise (HIP) -> game step + result (HIP; Othello's
       pass counter travels with the position, the trees forget it at every ply as the reference's do)
    -> finished games are replaced by fresh ones at once, their trees reset (HIP)
so every slot plays a live position on every ply.  Same search semantics as
BatchedMCTS.batch_playout; the schedule of temperatures follows game.py:55-63.

This file is what bench.py times.

With `record=True` the driver also keeps, in HBM, what the reference's harness keeps per ply
(game.py:97-108: position, visit distribution, root WDL, legal-move mask, side to move), moves
every finished game's rows into a store of finished games with one scatter per ply (no host
synchronisation), and `drain()` turns that store into the reference's exact `play_data`
tuples (game.py:110-157: winner_z, steps_to_end, aux targets, td_steps future root WDL and the
terminal tuple) - SURVEY section 8f row f1.  Two switches exist for parity tests against the
reference harness, which carries finished games to the end and samples with numpy:
`refill=False` leaves a finished slot dead instead of starting a new game in it, and
`sampler="reference"` draws the moves on the host with the reference's procedure and numpy's
global generator (player.py:348-371).  The noise-epsilon decay over the plies of a game
(game.py:87-91, `noise_steps`) is per game here (fixture G14 pins the lockstep case).
"""

import numpy as np
import torch

from src import fused as F
from src.MCTS_cpp import BatchedMCTS


class DeviceSelfPlay:
    def __init__(self, net, n_games, n_playout=200, vl_batch=4, c_init=1.4, c_base=None, alpha=0.3,
                 noise_epsilon=0.25, fpu_reduction=0.2, use_symmetry=True, mlh_slope=0.1, mlh_cap=0.2,
                 value_decay=1.0, temperature=1.0, temp_decay_moves=20, temp_endgame=0.0, seed=0,
                 reserve_slots=None, record=False, td_steps=0, refill=True, sampler="device",
                 max_finished_games=None, table_log2=0, game="Connect4", score_utility_factor=0.0, score_scale=8.0,
                 noise_steps=0, noise_eps_min=0.1):
        self.B = int(n_games)
        self.n_playout = int(n_playout)
        self.vl_batch = int(vl_batch)
        self.temperature, self.temp_decay_moves, self.temp_endgame = temperature, temp_decay_moves, temp_endgame
        if c_base is None:
            c_base = 5 * n_playout                  # server.py:135 (c_base_factor 5)
        assert game in ("Connect4", "Othello")
        self.game = game
        self.game_id = 0 if game == "Connect4" else 1                # AZ_GAME_*
        # the longest game in plies: 42 stones; Othello: 60 stones + passes (never two in a row
        # before the end, so < 120)
        self.MAX_PLIES = 42 if game == "Connect4" else 126
        self.search = BatchedMCTS(self.B, c_init=c_init, c_base=c_base, alpha=alpha, n_playout=n_playout,
                                  game_name=game, noise_epsilon=noise_epsilon,
                                  fpu_reduction=fpu_reduction, use_symmetry=use_symmetry,
                                  mlh_slope=mlh_slope, mlh_cap=mlh_cap, value_decay=value_decay,
                                  score_utility_factor=score_utility_factor, score_scale=score_scale)
        self.search.seed(seed)
        self.fused = F.FusedSearch(self.search, net)
        if table_log2:
            self.fused.enable_table(table_log2)         # the reference's cache_size (src/Cache.py), in HBM
        self.h = self.fused.h
        dev = self.fused.device
        self.device = dev
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(int(seed))
        z = dict(device=dev)
        # initial position (Connect4: empty; Othello.h:62-75: black = player +1 on (3,4) and (4,3))
        self.start_p1, self.start_p2 = (0, 0) if game == "Connect4" else ((1 << 28) | (1 << 35), (1 << 27) | (1 << 36))
        self.bb_p1 = torch.full((self.B,), self.start_p1, dtype=torch.int64, **z)
        self.bb_p2 = torch.full((self.B,), self.start_p2, dtype=torch.int64, **z)
        self.turn = torch.ones(self.B, dtype=torch.int32, **z)
        # the game's memory beside the stones (Othello: consecutive passes; Connect4: unused)
        self.aux = torch.zeros(self.B, dtype=torch.int32, **z)
        self.ply = torch.zeros(self.B, dtype=torch.int32, **z)
        self.counts = torch.zeros((self.B, self.search.action_size), dtype=torch.int32, **z)
        self.actions = torch.zeros(self.B, dtype=torch.int32, **z)
        self.done = torch.zeros(self.B, dtype=torch.uint8, **z)
        self.winner = torch.zeros(self.B, dtype=torch.int32, **z)
        # running totals kept on the device: positions, games, p1 wins, p2 wins, draws
        self.totals = torch.zeros(5, dtype=torch.int64, **z)
        if reserve_slots is None:
            # A re-rooting copies the kept subtree into the tree's other arena half (k_prune), so a tree
            # occupies what is reachable from its root: the carried subtree plus one ply's growth of at
            # most n_playout * actions records, not a whole game's worth.  Reserve six plies' worst-case
            # growth per half: a tree is compacted once it could not take two more plies where it is, i.e.
            # every few plies; the engine checks the bound every ply from the occupancy the prune kernel
            # reports and grows the arenas if a tree ever needs more (a device-wide stop and a copy).
            want = 6 * self.n_playout * (7 if game == "Connect4" else 33)      # 33: most legal moves an Othello position has
            free_bytes, _ = torch.cuda.mem_get_info(dev)
            reserve_slots = min(want, int(free_bytes // 2) // (self.B * 2 * 48))
        if reserve_slots and int(reserve_slots) > 4096:
            F.check(F.lib().az_mcts_reserve(self.h, int(reserve_slots)))
        assert sampler in ("device", "reference")
        self.sampler, self.refill, self.record, self.td_steps = sampler, bool(refill), bool(record), int(td_steps)
        self.dead = torch.zeros(self.B, dtype=torch.bool, **z)        # refill=False: slots whose game is over
        self.ar = torch.arange(self.B, **z)
        # constants the step needs, on the device once (a torch.tensor(scalar, device=...) per ply
        # is a pageable host-to-device copy, which waits for the stream's work)
        self._c_temp = torch.tensor(float(self.temperature), **z)
        self._c_temp_end = torch.tensor(float(self.temp_endgame), **z)
        self._c_B = torch.tensor(self.B, dtype=torch.int64, **z)
        # the host only enqueues; it may run at most `max_plies_ahead` plies ahead of the device
        self.max_plies_ahead = 1
        self._ply_events = []
        # noise-epsilon decay over the plies of a game (game.py:87-91 with AlphaZeroPlayer.noise_steps /
        # noise_eps_min, player.py:122,159-161): the reference moves ONE epsilon for a batch of games that
        # start together; here every game decays from its own first ply (az_mcts_dev_set_noise_epsilons)
        self.noise_steps, self.noise_eps_init, self.noise_eps_min = int(noise_steps), float(noise_epsilon), float(noise_eps_min)
        if self.noise_steps > 0:
            self.eps_tree = torch.full((self.B,), self.noise_eps_init, dtype=torch.float32, **z)
            F.lib().az_mcts_dev_set_noise_epsilons.argtypes = [F.C.c_void_p, F.C.c_void_p]
            F.check(F.lib().az_mcts_dev_set_noise_epsilons(self.h, self.eps_tree.data_ptr()))
        if self.record:
            A = self.search.action_size
            self.stats = torch.zeros((self.B, 6 + 8 * A), dtype=torch.float32, **z)
            T = self.MAX_PLIES + 2                                    # plies + end state + one scratch column
            self.G = int(max_finished_games or max(4 * self.B, 1024))
            def rows(n):
                return dict(bb1=torch.zeros((n, T), dtype=torch.int64, **z), bb2=torch.zeros((n, T), dtype=torch.int64, **z),
                            turn=torch.zeros((n, T), dtype=torch.int8, **z), prob=torch.zeros((n, T, A), dtype=torch.float32, **z),
                            wdl=torch.zeros((n, T, 3), dtype=torch.float32, **z), mask=torch.zeros((n, T, A), dtype=torch.bool, **z))
            self.rec = rows(self.B)                                   # the games in progress
            self.fin = rows(self.G + 1)                               # finished games (+ one scratch row)
            self.fin_len = torch.zeros(self.G + 1, dtype=torch.int32, **z)
            self.fin_winner = torch.zeros(self.G + 1, dtype=torch.int32, **z)
            self.fin_slot = torch.zeros(self.G + 1, dtype=torch.int32, **z)
            self.n_finished = torch.zeros((), dtype=torch.int64, **z)
            self.n_dropped = torch.zeros((), dtype=torch.int64, **z)
            self.mask_u8 = torch.zeros((self.B, A), dtype=torch.uint8, **z)

    def _pick_actions(self):
        """Visit counts -> move: proportional to N^(1/T) while T > 0, arg-max otherwise
        (player.py:348-371 with the temperature schedule of game.py:55-63)."""
        visits = self.counts.to(torch.float32)
        greedy = visits.argmax(dim=1)
        if self.temp_decay_moves <= 0:
            temps = torch.full((self.B,), float(self.temperature), device=self.device)
        else:
            temps = torch.where(self.ply < self.temp_decay_moves, self._c_temp, self._c_temp_end)
        hot = temps > 1e-6
        t_eff = torch.where(hot, temps, torch.ones_like(temps))
        w = visits.clamp_min(0).pow(1.0 / t_eff.unsqueeze(1))
        w = torch.where(visits > 0, w, torch.zeros_like(w))
        safe = torch.where(w.sum(1, keepdim=True) > 0, w, torch.ones_like(w))
        sampled = torch.multinomial(safe, 1, generator=self.gen).squeeze(1)
        self.actions.copy_(torch.where(hot, sampled, greedy).to(torch.int32))

    def _pick_actions_reference(self):
        """player.py:348-371 verbatim on the host (numpy's global generator): for parity runs."""
        visits = self.counts.cpu().numpy().astype(np.int64)
        ply = self.ply.cpu().numpy()
        dead = self.dead.cpu().numpy()
        acts = np.zeros(self.B, dtype=np.int32)
        for i in range(self.B):
            visit = visits[i]
            valid = visit > 0
            if dead[i] or not valid.any():
                acts[i] = -1 if dead[i] else 0
                continue
            if self.temp_decay_moves <= 0:
                temp = self.temperature
            else:
                temp = self.temperature if ply[i] < self.temp_decay_moves else self.temp_endgame
            if temp <= 1e-6:
                acts[i] = int(np.argmax(visit))
            else:
                log_visits = np.log(visit[valid])
                x = log_visits / temp
                pr = np.exp(x - np.max(x))
                acts[i] = int(np.random.choice(np.where(valid)[0], p=pr / np.sum(pr)))
        self.actions.copy_(torch.from_numpy(acts).to(self.device))

    def _record_position(self):
        """What game.py:97-108 appends before the move is played."""
        idx = torch.where(self.dead, torch.full_like(self.ply, self.MAX_PLIES + 1), self.ply).long()
        r = self.rec
        r["bb1"][self.ar, idx] = self.bb_p1
        r["bb2"][self.ar, idx] = self.bb_p2
        r["turn"][self.ar, idx] = self.turn.to(torch.int8)
        visits = self.counts.to(torch.float64)                      # player.py:356: int / int in double, stored as f32
        tot = visits.sum(1, keepdim=True)
        r["prob"][self.ar, idx] = torch.where(tot > 0, visits / tot.clamp_min(1), torch.zeros_like(visits)).to(torch.float32)
        r["wdl"][self.ar, idx] = self.stats[:, 3:6]
        F.check(F.lib().az_game_dev_valid_mask(self.game_id, self.bb_p1.data_ptr(), self.bb_p2.data_ptr(), self.turn.data_ptr(),
                                               self.aux.data_ptr(), self.mask_u8.data_ptr(), self.B, F._stream()))
        r["mask"][self.ar, idx] = self.mask_u8.bool()

    def _record_finished(self, fin):
        """End state of the games that just finished, then their rows move to the finished store."""
        idx = torch.where(fin, self.ply, torch.full_like(self.ply, self.MAX_PLIES + 1)).long()   # ply already counts the last move
        r = self.rec
        r["bb1"][self.ar, idx] = self.bb_p1
        r["bb2"][self.ar, idx] = self.bb_p2
        r["turn"][self.ar, idx] = self.turn.to(torch.int8)
        rank = torch.cumsum(fin.to(torch.int64), 0) - 1
        dst = self.n_finished + rank
        keep = fin & (dst < self.G)
        dst = torch.where(keep, dst, torch.full_like(dst, self.G))
        for k, v in self.fin.items():
            v.index_copy_(0, dst, r[k])
        self.fin_len.index_copy_(0, dst, self.ply)
        self.fin_winner.index_copy_(0, dst, self.winner)
        self.fin_slot.index_copy_(0, dst, self.ar.to(torch.int32))
        self.n_finished += keep.sum()
        self.n_dropped += (fin & ~keep).sum()

    def step(self):
        """One ply in every game."""
        L = F.lib()
        s = F._stream()
        F.check(L.az_mcts_dev_set_roots(self.h, self.bb_p1.data_ptr(), self.bb_p2.data_ptr(),
                                        self.turn.data_ptr(), s))
        if self.noise_steps > 0:
            # in double, as the reference's Python arithmetic, then one rounding to the config's float
            decay = (1.0 - self.ply.double() / self.noise_steps).clamp_min(0.0)
            self.eps_tree.copy_((self.noise_eps_min + (self.noise_eps_init - self.noise_eps_min) * decay).float())
        self.fused.search(self.n_playout, self.vl_batch)
        F.check(L.az_mcts_dev_counts(self.h, self.counts.data_ptr(), s))
        if self.sampler == "reference":
            self._pick_actions_reference()
        else:
            self._pick_actions()
            if not self.refill:
                self.actions.masked_fill_(self.dead, -1)
        if self.record:
            F.check(L.az_mcts_dev_root_stats(self.h, self.stats.data_ptr(), s))
            self._record_position()
        F.check(L.az_mcts_dev_prune_roots(self.h, self.actions.data_ptr(), s))
        # with recording the end state has to survive the step: finished boards are reset below
        kernel_refill = 1 if (self.refill and not self.record) else 0
        F.check(L.az_game_dev_step(self.game_id, self.bb_p1.data_ptr(), self.bb_p2.data_ptr(), self.turn.data_ptr(),
                                   self.aux.data_ptr(), self.actions.data_ptr(), self.done.data_ptr(),
                                   self.winner.data_ptr(), self.B, kernel_refill, s))
        F.check(L.az_mcts_dev_reset_masked(self.h, self.done.data_ptr(), s))
        fin_b = self.done.bool()
        fin = self.done.to(torch.int64)
        self.ply = torch.where(self.dead, self.ply, self.ply + 1)
        if self.record:
            self._record_finished(fin_b)
        if self.refill:
            if not kernel_refill:
                self.bb_p1.masked_fill_(fin_b, self.start_p1)
                self.bb_p2.masked_fill_(fin_b, self.start_p2)
                self.turn.masked_fill_(fin_b, 1)
                self.aux.masked_fill_(fin_b, 0)
            self.ply = torch.where(fin_b, torch.zeros_like(self.ply), self.ply)
        else:
            self.dead |= fin_b
        self.totals += torch.stack([self._c_B, fin.sum(),
                                    (fin * (self.winner == 1)).sum(), (fin * (self.winner == -1)).sum(),
                                    (fin * (self.winner == 0)).sum()])
        # sticky device error word (arena full, compact list overflow), polled without a stall:
        # raises one ply after the fact instead of letting the trees thin out silently
        F.check(L.az_mcts_dev_check(self.h, s))
        # bounded run-ahead: ~550 launches per ply would otherwise pile up without limit
        ev = torch.cuda.Event()
        ev.record()
        self._ply_events.append(ev)
        if len(self._ply_events) > self.max_plies_ahead:
            self._ply_events.pop(0).synchronize()

    def drain(self):
        """Finished games since the last call, as the reference's `batch_self_play` returns them:
        a list of (winner, play_data) with play_data the tuple of per-ply tuples of
        game.py:131-157 (+ the terminal tuple), plus the slot each game was played in.
        Synchronises; the store is emptied."""
        assert self.record
        n = int(self.n_finished.item())
        host = {k: v[:n].cpu().numpy() for k, v in self.fin.items()}
        lens = self.fin_len[:n].cpu().numpy()
        winners = self.fin_winner[:n].cpu().numpy()
        slots = self.fin_slot[:n].cpu().numpy()
        self.n_finished.zero_()
        games = []
        k = self.td_steps
        for g in range(n):
            T = int(lens[g])
            winner = int(winners[g])
            states = planes_from_bitboards(host["bb1"][g, :T + 1], host["bb2"][g, :T + 1], host["turn"][g, :T + 1], self.game)
            winner_z = np.full(T, winner, dtype=np.int32)
            steps_to_end = np.arange(T, 0, -1, dtype=np.int32)
            if self.game == "Othello":                               # game.py:17-30: final disc difference, mover's view
                diff = int(bin(int(host["bb1"][g, T]) & (2 ** 64 - 1)).count("1")) - int(bin(int(host["bb2"][g, T]) & (2 ** 64 - 1)).count("1"))
                aux = diff * np.asarray(host["turn"][g, :T], dtype=np.int32)
                terminal_aux = diff * int(host["turn"][g, T])
            else:
                aux = steps_to_end                                   # Connect4: moves left
                terminal_aux = 0
            probs, masks = host["prob"][g], host["mask"][g]
            # object identities as in game.py:121-157 (one root-WDL object per ply, reused by the
            # td-step column; one zero vector per game): pickle writes shared objects once, so
            # the upload below is byte-identical to the reference client's only if they match
            root_wdls = [host["wdl"][g][t] for t in range(T)]
            zero_wdl = np.zeros(3, dtype=np.float32)
            cols = [[states[t] for t in range(T)], [probs[t] for t in range(T)], winner_z, steps_to_end, aux,
                    root_wdls, [masks[t] for t in range(T)]]
            if k > 0:
                cols.append([root_wdls[t + k] if t + k < T else zero_wdl for t in range(T)])
            play = list(zip(*cols))
            terminal = [states[T], np.zeros_like(probs[0]), winner, 0, terminal_aux, zero_wdl, np.ones_like(masks[0])]
            if k > 0:
                terminal.append(zero_wdl)
            play.append(tuple(terminal))
            games.append((winner, tuple(play), int(slots[g])))
        return games

    def read_totals(self):
        t = self.totals.cpu().tolist()          # synchronises
        return dict(positions=t[0], games=t[1], p1_wins=t[2], p2_wins=t[3], draws=t[4])

    def engine_counters(self):
        return F.counters(self.h)


class StreamedSelfPlay:
    """`n_games` games as `streams` independent DeviceSelfPlay drivers, each with its own engine,
    HIP stream and host thread.

    One driver's selection and backup kernels keep one wavefront per SIMD busy and end when the
    deepest tree of the batch is done - most of the chip idles under them; the evaluator kernels
    are issue bound and fill it.  Games never interact (the reference runs them in separate
    OpenMP iterations and batches them only for the network, BatchedMCTS.h:88-135), so the batch
    can be cut into groups whose iterations overlap on the device: one group's tree kernels run
    under another group's evaluator.  Each game sees exactly the search it would see in a single
    driver (the evaluator computes every leaf independently of its batch); what changes is which
    slots share a batch and the device generator's seeds (driver i uses seed * streams + i).

    `step(n)` plays n plies in every game: the drivers' host threads only enqueue work (ctypes and
    torch release the GIL while they do) and are joined, the device is not waited for - call
    `synchronize()` for that.  `drain()`, `read_totals()`, `engine_counters()` aggregate over the
    drivers; slot numbers are global (driver offset + local slot)."""

    def __init__(self, net, n_games, streams=2, seed=0, **kw):
        from concurrent.futures import ThreadPoolExecutor
        p = next(net.parameters(), None)
        self.device = p.device if p is not None else torch.device("cuda", torch.cuda.current_device())
        # a process has four hardware queues on ROCm and the NULL stream owns one: beyond three
        # drivers, streams share a queue and gain little
        assert 1 <= int(streams) <= 8, "StreamedSelfPlay: 1 to 8 streams"
        streams = max(1, min(int(streams), int(n_games)))
        base, extra = divmod(int(n_games), streams)
        self.sizes = [base + (1 if i < extra else 0) for i in range(streams)]
        self.offsets = [sum(self.sizes[:i]) for i in range(streams)]
        self.B = int(n_games)
        self.streams = [torch.cuda.Stream(self.device) for _ in range(streams)]
        self.parts = []
        for i, st in enumerate(self.streams):
            with torch.cuda.stream(st):
                self.parts.append(DeviceSelfPlay(net, self.sizes[i], seed=int(seed) * streams + i, **kw))
        self.synchronize()
        self._pool = ThreadPoolExecutor(max_workers=streams, thread_name_prefix="az-selfplay")

    def _run(self, i, n):
        with torch.cuda.device(self.device), torch.cuda.stream(self.streams[i]):
            for _ in range(n):
                self.parts[i].step()

    def step(self, n=1):
        if len(self.parts) == 1:
            return self._run(0, n)
        for f in [self._pool.submit(self._run, i, n) for i in range(len(self.parts))]:
            f.result()

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def drain(self):
        self.synchronize()
        games = []
        for off, part, st in zip(self.offsets, self.parts, self.streams):
            with torch.cuda.stream(st):
                games += [(w, play, slot + off) for (w, play, slot) in part.drain()]
        return games

    def read_totals(self):
        self.synchronize()
        out = {}
        for part, st in zip(self.parts, self.streams):
            with torch.cuda.stream(st):
                for k, v in part.read_totals().items():
                    out[k] = out.get(k, 0) + v
        return out

    def engine_counters(self):
        self.synchronize()
        out = {}
        for part in self.parts:
            for k, v in part.engine_counters().items():
                out[k] = out.get(k, 0) + v
        return out

    def table_stats(self):
        self.synchronize()
        out = {}
        for part in self.parts:
            for k, v in part.fused.table_stats().items():
                out[k] = out.get(k, 0) + v
        if out.get("lookups"):
            out["hit_rate"] = out["hits"] / out["lookups"]
        return out

    def close(self):
        self._pool.shutdown(wait=True)


def pack_upload(games):
    """The body of the actor's POST /upload for these games (client.py:366-368 with 386-388):
    pickle of {'__az__': True, 'data': [play_data, ...]} at the highest protocol.  `games` is
    what `DeviceSelfPlay.drain()` (or the reference's `batch_self_play`) returned."""
    import pickle
    return pickle.dumps({'__az__': True, 'data': [g[1] for g in games]}, protocol=pickle.HIGHEST_PROTOCOL)


def planes_from_bitboards(bb_p1, bb_p2, turn, game="Connect4"):
    """(n,) bitboards and side to move -> the (n, 3, rows, cols) int8 planes of `Env.current_state()`
    (env_common.h:93-119): stones of the side to move, stones of the opponent, the turn sign
    everywhere.  Connect4: bit 7*col + height, row 0 is the top of the board; Othello: bit 8*row + col."""
    bb_p1 = np.asarray(bb_p1).astype(np.uint64)
    bb_p2 = np.asarray(bb_p2).astype(np.uint64)
    turn = np.asarray(turn).astype(np.int8)
    n = bb_p1.shape[0]
    own = np.where(turn > 0, bb_p1, bb_p2)
    opp = np.where(turn > 0, bb_p2, bb_p1)
    if game == "Othello":
        out = np.zeros((n, 3, 8, 8), dtype=np.int8)
        for r in range(8):
            for c in range(8):
                bit = np.uint64(8 * r + c)
                out[:, 0, r, c] = (own >> bit) & np.uint64(1)
                out[:, 1, r, c] = (opp >> bit) & np.uint64(1)
    else:
        out = np.zeros((n, 3, 6, 7), dtype=np.int8)
        for c in range(7):
            for h in range(6):
                bit = np.uint64(7 * c + h)
                out[:, 0, 5 - h, c] = (own >> bit) & np.uint64(1)
                out[:, 1, 5 - h, c] = (opp >> bit) & np.uint64(1)
    out[:, 2] = turn[:, None, None]
    return out
