"""Long run of the device self-play driver at the benchmark shape (GPU box only):
    python tools/soak_selfplay.py [steps] [table_log2]
8192 games x `steps` plies with trajectory recording (and optionally the transposition table);
prints throughput per 10 plies, drains the finished games and checks their invariants."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero-al_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from src.az_net import Connect4Net  # noqa: E402
from src.selfplay import DeviceSelfPlay  # noqa: E402
import scenarios as S  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
table = int(sys.argv[2]) if len(sys.argv) > 2 else 0
torch.manual_seed(0)
net = Connect4Net(device="cuda").eval()
sp = DeviceSelfPlay(net, 8192, n_playout=200, vl_batch=4, seed=0, record=True, td_steps=2, table_log2=table,
                    max_finished_games=65536)
t0 = time.perf_counter()
tl = t0
for i in range(steps):
    sp.step()
    if (i + 1) % 10 == 0:
        torch.cuda.synchronize()
        now = time.perf_counter()
        print("plies %3d: %.0f positions/s over the last 10" % (i + 1, 10 * 8192 / (now - tl)), flush=True)
        tl = now
torch.cuda.synchronize()
el = time.perf_counter() - t0
tot = sp.read_totals()
games = sp.drain()
print("total: %d positions in %.1f s = %.0f positions/s; %d games finished (p1 %d, p2 %d, draws %d), dropped %d"
      % (tot["positions"], el, tot["positions"] / el, tot["games"], tot["p1_wins"], tot["p2_wins"], tot["draws"],
         int(sp.n_dropped.item())))
if table:
    print("table:", sp.fused.table_stats())
assert len(games) == tot["games"]
lens = []
for winner, play, slot in games[:2000]:
    T = len(play) - 1
    lens.append(T)
    end = play[-1][0]
    to_move = 1 if T % 2 == 0 else -1
    board = end[0].astype(np.int8) * to_move - end[1].astype(np.int8) * to_move
    assert S.np_winner(board) == winner and S.np_done(board), (slot, T)
    for t in (0, T // 2, T - 1):
        st, prob, z, stl, aux, wdl, mask, fut = play[t]
        assert int(st[:2].sum()) == t and abs(float(prob.sum()) - 1) < 1e-5 and stl == T - t and z == winner
print("game length: mean %.1f, min %d, max %d (first %d games checked)" % (np.mean(lens), min(lens), max(lens), len(lens)))
c = sp.engine_counters()
print("engine counters:", c)
