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
    sp = StreamedSelfPlay(net, games, streams=P, n_playout=200, vl_batch=4, reserve_slots=49152)
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
    assert tot["positions"] == plies * games and cnt["sims"] == plies * games * 200, (tot, cnt)
    print("ok", cnt, flush=True)


if __name__ == "__main__":
    main()
