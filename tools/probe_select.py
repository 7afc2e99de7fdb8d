"""Tree kernels alone at the benchmark shape: continuous self-play of 8192 Connect4 games with the
native integer-hash evaluator (no network), selection / backup kernels timed by HIP events on every
launch, plus a checksum of what was played (identical across kernel variants: the device generator
is counter-based, so two builds that compute the same search play the same games).

    python tools/probe_select.py [games] [plies] [lead_in] [n_playout] [K]
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "alphazero-al_amd"), ROOT]

import torch  # noqa: E402


def main():
    games = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    plies = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    lead = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 200
    K = int(sys.argv[5]) if len(sys.argv) > 5 else 4
    from src import fused as F
    from src.hash_eval import HashEvaluator
    from src.selfplay import DeviceSelfPlay
    sp = DeviceSelfPlay(HashEvaluator("cuda"), games, n_playout=n, vl_batch=K, seed=0)
    L = F.lib()
    for _ in range(lead):
        sp.step()
    torch.cuda.synchronize()
    F.check(L.az_mcts_counters_reset(sp.h))
    F.check(L.az_mcts_profile(sp.h, 1))
    t0 = time.perf_counter()
    done = 0
    while done < plies:                      # the event rings hold 8192 launches: read them every 40 plies
        chunk = min(40, plies - done)
        for _ in range(chunk):
            sp.step()
        done += chunk
        torch.cuda.synchronize()
        if done == chunk:
            ms = (C.c_double * 2)(); nl = (C.c_int64 * 2)()
            F.check(L.az_mcts_profile_read(sp.h, C.byref(ms), C.byref(nl)))
            F.check(L.az_mcts_profile(sp.h, 0))
    el = time.perf_counter() - t0
    cnt = sp.engine_counters()
    tot = sp.read_totals()
    sig = int(sp.bb_p1.sum().item()) ^ (int(sp.bb_p2.sum().item()) << 1)
    print("variant=%s games=%d plies=%d: select %.2f us  backprop %.2f us  (%d / %d launches, event pairs included), "
          "%.2f ms/ply, levels/sim %.3f, exp/sim %.3f, totals %s sig %x"
          % (os.environ.get("AZ_SELECT_VARIANT", "default"), games, plies, ms[0] / max(nl[0], 1) * 1e3, ms[1] / max(nl[1], 1) * 1e3,
             nl[0], nl[1], el / plies * 1e3, cnt["levels"] / max(cnt["sims"], 1), cnt["expansions"] / max(cnt["sims"], 1),
             {k: tot[k] for k in ("games", "p1_wins", "p2_wins", "draws")}, sig & 0xffffffffffff), flush=True)


if __name__ == "__main__":
    main()
