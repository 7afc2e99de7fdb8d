"""What an UNCHANGED caller gets: the reference's per-ply Python driver (game.py:65-164 with player.py:333-375, restated in
tests/harness.py) over our drop-in modules - `env_cpp.connect4.Env` objects on the host, `BatchedMCTS.batch_playout` with the
network as a torch module on the GPU (which takes the fused device loop underneath) - against `DeviceSelfPlay`, which keeps
games and trees in HBM.  GPU box only.

    python tools/measure_dropin.py [games] [plies]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "alphazero-al_amd"), ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402
import harness  # noqa: E402
from src import MCTS_cpp  # noqa: E402
from src.az_net import Connect4Net  # noqa: E402
from src.env_cpp.connect4 import Env  # noqa: E402
from src.selfplay import DeviceSelfPlay  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
plies = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n, K = 200, 4
torch.manual_seed(1234)
net = Connect4Net(device="cuda").eval()
w = MCTS_cpp.BatchedMCTS(B, c_init=1.4, c_base=5 * n, alpha=0.3, n_playout=n, noise_epsilon=0.25, fpu_reduction=0.2,
                         use_symmetry=True, mlh_slope=0.1, mlh_cap=0.2)
w.seed(0)
envs = [Env() for _ in range(B)]
np.random.seed(0)
parts = dict(boards=0.0, search=0.0, read=0.0, choose=0.0, prune=0.0, step=0.0)
t_all = time.perf_counter()
done_games = 0
for ply in range(plies + 1):
    t0 = time.perf_counter()
    boards = np.array([e.board for e in envs])
    turns = np.array([e.turn for e in envs], dtype=np.int32)
    t1 = time.perf_counter()
    w.batch_playout(net, boards, turns, vl_batch=K)
    t2 = time.perf_counter()
    visits = w.get_visits_count()
    rs = w.get_root_stats()
    t3 = time.perf_counter()
    acts, dists = harness._choose(visits, [1.0] * B, w.action_size)
    t4 = time.perf_counter()
    w.prune_roots(np.array(acts, dtype=np.int32))
    t5 = time.perf_counter()
    for i, e in enumerate(envs):
        e.step(acts[i])
        if e.done():
            done_games += 1
            e.reset()
            w.reset_env(i)
    t6 = time.perf_counter()
    if ply == 0:                     # the first ply pays the one-time costs (kernel selection, allocations)
        t_all = time.perf_counter()
        continue
    for k, d in zip(parts, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
        parts[k] += d
dt = time.perf_counter() - t_all
drop = dict(positions_per_s=round(B * plies / dt, 1), ms_per_ply=round(dt / plies * 1e3, 1),
            ms_per_ply_by_part={k: round(v / plies * 1e3, 1) for k, v in parts.items()}, games_finished=done_games)
del w
sp = DeviceSelfPlay(net, B, n_playout=n, vl_batch=K, seed=0)
sp.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(plies):
    sp.step()
torch.cuda.synchronize()
dt2 = time.perf_counter() - t0
print(json.dumps(dict(games=B, plies=plies, n_playout=n, vl_batch=K,
                      unchanged_caller=drop, device_selfplay=dict(positions_per_s=round(B * plies / dt2, 1), ms_per_ply=round(dt2 / plies * 1e3, 1)))))
