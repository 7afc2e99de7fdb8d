"""Long run of the streamed driver (selfplay.StreamedSelfPlay): `plies` plies of `games` games on
`streams` streams, a progress line every 100 plies.

    python tools/soak_streams.py [streams] [plies] [games]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "alphazero-al_amd"), ROOT]

import torch  # noqa: E402
from src.az_net import Connect4Net  # noqa: E402
from src.selfplay import StreamedSelfPlay  # noqa: E402


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    plies = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    games = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
    torch.manual_seed(1234)
    net = Connect4Net(device=torch.device("cuda", 0)).eval()
    n_playout = int(os.environ.get("N_PLAYOUT", "200"))
    if os.environ.get("EVALUATOR") == "hash":
        from src.hash_eval import HashEvaluator
        net = HashEvaluator("cuda")
    # default reservation (six plies' growth per arena half): the run shows that no tree outgrows it
    sp = StreamedSelfPlay(net, games, streams=P, n_playout=n_playout, vl_batch=4)
    cap0 = [int(__import__("src.fused", fromlist=["lib"]).lib().az_mcts_capacity(p.h)) for p in sp.parts]
    t0 = time.perf_counter()
    done = 0
    while done < plies:
        n = min(100, plies - done)
        sp.step(n)
        sp.synchronize()
        done += n
        tot = sp.read_totals()
        el = time.perf_counter() - t0
        print(f"streams={P} plies={done} positions={tot['positions']} games={tot['games']} "
              f"({tot['p1_wins']}/{tot['p2_wins']}/{tot['draws']}) {tot['positions'] / el:,.0f} positions/s", flush=True)
    cnt = sp.engine_counters()
    assert tot["positions"] == plies * games and cnt["sims"] == plies * games * n_playout, (tot, cnt)
    from src import fused as F
    cap1 = [int(F.lib().az_mcts_capacity(p.h)) for p in sp.parts]
    used = []
    for p in sp.parts:
        v = F.C.c_int64()
        F.check(F.lib().az_mcts_max_used(p.h, F.C.byref(v)))
        used.append(v.value)
    print("ok", cnt, "arena records per half: reserved", cap0, "now", cap1, "fullest tree now", used, flush=True)


if __name__ == "__main__":
    main()
