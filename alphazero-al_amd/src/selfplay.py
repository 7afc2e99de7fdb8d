"""Device-side self-play driver: thousands of Connect4 games and their search trees advance
in lockstep without leaving HBM.

The reference's driver (src/game.py:65-164 with player.py:333-375) is O(batch) Python per ply
- it rebuilds the numpy boards from per-game Env objects, samples each action in a Python
loop and keeps finished games in the batch until the slowest one ends.  Here a ply is
    roots (bitboards in HBM) -> FusedSearch.search (n_playout simulations per tree)
    -> root visit counts (HIP) -> temperature sampling (torch, on device)
    -> prune_roots with fresh device Dirichlet noise (HIP) -> game step + result (HIP)
    -> finished games are replaced by fresh ones at once, their trees reset (HIP)
so every slot plays a live position on every ply.  Same search semantics as
BatchedMCTS.batch_playout; the schedule of temperatures follows game.py:55-63.

This file is what bench.py times.  Recording full training trajectories
(game.py:131-157 tuples) on the device is the next step.
"""
import ctypes as C

import torch

from src import fused as F
from src.MCTS_cpp import BatchedMCTS


class DeviceSelfPlay:
    def __init__(self, net, n_games, n_playout=200, vl_batch=4, c_init=1.4, c_base=None, alpha=0.3,
                 noise_epsilon=0.25, fpu_reduction=0.2, use_symmetry=True, mlh_slope=0.1, mlh_cap=0.2,
                 value_decay=1.0, temperature=1.0, temp_decay_moves=20, temp_endgame=0.0, seed=0,
                 reserve_slots=None):
        self.B = int(n_games)
        self.n_playout = int(n_playout)
        self.vl_batch = int(vl_batch)
        self.temperature, self.temp_decay_moves, self.temp_endgame = temperature, temp_decay_moves, temp_endgame
        if c_base is None:
            c_base = 5 * n_playout                  # server.py:135 (c_base_factor 5)
        self.search = BatchedMCTS(self.B, c_init=c_init, c_base=c_base, alpha=alpha, n_playout=n_playout,
                                  game_name='Connect4', noise_epsilon=noise_epsilon,
                                  fpu_reduction=fpu_reduction, use_symmetry=use_symmetry,
                                  mlh_slope=mlh_slope, mlh_cap=mlh_cap, value_decay=value_decay)
        self.search.seed(seed)
        self.fused = F.FusedSearch(self.search, net)
        self.h = self.fused.h
        dev = self.fused.device
        self.device = dev
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(int(seed))
        z = dict(device=dev)
        self.bb_p1 = torch.zeros(self.B, dtype=torch.int64, **z)
        self.bb_p2 = torch.zeros(self.B, dtype=torch.int64, **z)
        self.turn = torch.ones(self.B, dtype=torch.int32, **z)
        self.ply = torch.zeros(self.B, dtype=torch.int32, **z)
        self.counts = torch.zeros((self.B, self.search.action_size), dtype=torch.int32, **z)
        self.actions = torch.zeros(self.B, dtype=torch.int32, **z)
        self.done = torch.zeros(self.B, dtype=torch.uint8, **z)
        self.winner = torch.zeros(self.B, dtype=torch.int32, **z)
        # running totals kept on the device: positions, games, p1 wins, p2 wins, draws
        self.totals = torch.zeros(5, dtype=torch.int64, **z)
        if reserve_slots:
            F.check(F.lib().az_mcts_reserve(self.h, int(reserve_slots)))

    def _pick_actions(self):
        """Visit counts -> move: proportional to N^(1/T) while T > 0, arg-max otherwise
        (player.py:348-371 with the temperature schedule of game.py:55-63)."""
        visits = self.counts.to(torch.float32)
        greedy = visits.argmax(dim=1)
        if self.temp_decay_moves <= 0:
            temps = torch.full((self.B,), float(self.temperature), device=self.device)
        else:
            temps = torch.where(self.ply < self.temp_decay_moves,
                                torch.tensor(float(self.temperature), device=self.device),
                                torch.tensor(float(self.temp_endgame), device=self.device))
        hot = temps > 1e-6
        t_eff = torch.where(hot, temps, torch.ones_like(temps))
        w = visits.clamp_min(0).pow(1.0 / t_eff.unsqueeze(1))
        w = torch.where(visits > 0, w, torch.zeros_like(w))
        safe = torch.where(w.sum(1, keepdim=True) > 0, w, torch.ones_like(w))
        sampled = torch.multinomial(safe, 1, generator=self.gen).squeeze(1)
        self.actions.copy_(torch.where(hot, sampled, greedy).to(torch.int32))

    def step(self):
        """One ply in every game."""
        L = F.lib()
        s = F._stream()
        F.check(L.az_mcts_dev_set_roots(self.h, self.bb_p1.data_ptr(), self.bb_p2.data_ptr(),
                                        self.turn.data_ptr(), s))
        self.fused.search(self.n_playout, self.vl_batch)
        F.check(L.az_mcts_dev_counts(self.h, self.counts.data_ptr(), s))
        self._pick_actions()
        F.check(L.az_mcts_dev_prune_roots(self.h, self.actions.data_ptr(), s))
        F.check(L.az_c4_dev_step(self.bb_p1.data_ptr(), self.bb_p2.data_ptr(), self.turn.data_ptr(),
                                 self.actions.data_ptr(), self.done.data_ptr(), self.winner.data_ptr(),
                                 self.B, 1, s))
        F.check(L.az_mcts_dev_reset_masked(self.h, self.done.data_ptr(), s))
        fin = self.done.to(torch.int64)
        self.ply = torch.where(self.done.bool(), torch.zeros_like(self.ply), self.ply + 1)
        self.totals += torch.stack([torch.tensor(self.B, device=self.device), fin.sum(),
                                    (fin * (self.winner == 1)).sum(), (fin * (self.winner == -1)).sum(),
                                    (fin * (self.winner == 0)).sum()])

    def read_totals(self):
        t = self.totals.cpu().tolist()          # synchronises
        return dict(positions=t[0], games=t[1], p1_wins=t[2], p2_wins=t[3], draws=t[4])

    def engine_counters(self):
        return F.counters(self.h)
