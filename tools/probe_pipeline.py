"""Experiment: the benchmark's 8192 games as P independent drivers of 8192/P games, each on
its own HIP stream and host thread, so that one group's tree kernels (latency bound, one
wavefront per SIMD, long tail) run under another group's evaluator kernels.

    python tools/probe_pipeline.py [P] [games] [steps] [lead_in]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "alphazero-al_amd"), ROOT]

import torch  # noqa: E402
from src.az_net import Connect4Net  # noqa: E402


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    games = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    lead = int(sys.argv[4]) if len(sys.argv) > 4 else 12
    global JOIN
    JOIN = int(os.environ.get('JOIN', '1'))
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    net = Connect4Net(device=dev).eval()
    from src.selfplay import StreamedSelfPlay
    sp = StreamedSelfPlay(net, games, streams=P, n_playout=200, vl_batch=4)

    def phase(n):
        sp.step(n) if JOIN == 0 else [sp.step() for _ in range(n)]
        torch.cuda.synchronize()

    phase(lead)
    phase(2)
    t0 = time.perf_counter()
    phase(steps)
    el = time.perf_counter() - t0
    tot = sp.read_totals()
    print(f"P={P} JOIN={JOIN} finished={tot['games']} games={games} steps={steps}: {games * steps / el:,.0f} positions/s, {el / steps * 1e3:.2f} ms/step",
          flush=True)


if __name__ == "__main__":
    main()
