"""BASELINE config 3: Othello, 4096 games, n_playout 400, virtual-loss batch 4,
score utility 0.15, with the reference's Othello network (az_net.OthelloNet, h_dim 256, random
init) as the evaluator.  The tree kernels and the games are the HIP engine's; the network runs
through its HIP twin (fast_othello.py: the 3x3 convolutions on nn_othello.hip) or, with
AZ_FUSED_FASTNET=0, as the torch module under bf16 autocast (library kernels).  EVALUATOR=hash
swaps in the integer-hash evaluator (tree kernels only).  Prints one JSON line.

    python tools/bench_othello.py [games] [n_playout] [timed_plies] [lead_in]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "alphazero-al_amd"), ROOT]

import torch  # noqa: E402
from src.az_net import OthelloNet  # noqa: E402
from src.selfplay import DeviceSelfPlay  # noqa: E402


def main():
    games = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    n_playout = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    plies = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    lead = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    evaluator = os.environ.get("EVALUATOR", "cnn")
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    torch.backends.cudnn.benchmark = True
    if evaluator == "hash":
        from src.hash_eval import OthelloHashEvaluator
        net = OthelloHashEvaluator(dev)
    else:
        net = OthelloNet(device=dev).to(memory_format=torch.channels_last)
    sp = DeviceSelfPlay(net, games, n_playout=n_playout, vl_batch=4, game="Othello", score_utility_factor=0.15,
                        score_scale=8.0, seed=0,
                        reserve_slots=int(os.environ["AZ_RESERVE_SLOTS"]) if "AZ_RESERVE_SLOTS" in os.environ else None,
                        table_log2=int(os.environ.get("TABLE_LOG2", "0")))
    for i in range(lead):
        t = time.perf_counter()
        sp.step()
        torch.cuda.synchronize()
        print(f"lead-in ply {i + 1}/{lead}: {time.perf_counter() - t:.2f} s", file=sys.stderr, flush=True)
    c0 = sp.engine_counters()
    t0 = time.perf_counter()
    for i in range(plies):
        sp.step()
        torch.cuda.synchronize()
        print(f"timed ply {i + 1}/{plies}: {time.perf_counter() - t0:.2f} s", file=sys.stderr, flush=True)
    el = time.perf_counter() - t0
    c1 = sp.engine_counters()
    sims = c1["sims"] - c0["sims"]
    print(json.dumps({"metric": "self-play positions/sec (Othello n_playout=%d, %d games, vl_batch=4, score utility 0.15)" % (n_playout, games),
                      "value": round(games * plies / el, 1), "unit": "positions/s", "ms_per_step": round(el / plies * 1e3, 1),
                      "sims_per_s": round(sims / el, 1), "node_expansions_per_s": round((c1["expansions"] - c0["expansions"]) / el, 1),
                      "mean_select_depth": round((c1["levels"] - c0["levels"]) / max(sims, 1), 3),
                      "evaluator": ("integer-hash evaluator" if evaluator == "hash" else
                                    "OthelloNet h_dim 256, 3 residual blocks, random init, " +
                                    (("HIP twin as one native model object (nn_othello.hip, nn_othello_heads.hip; native loop)"
                                      if sp.fused._native_model() is not None else "HIP twin (nn_othello.hip convolutions, Python loop)")
                                     if type(sp.fused.fast).__name__ == "FastOthelloNet"
                                     else "torch module under bf16 autocast (library kernels)")),
                      "timed_plies": plies, "lead_in_plies": lead,
                      "transposition_table": (sp.fused.table_stats() if sp.fused.table_log2 else None)}), flush=True)


if __name__ == "__main__":
    main()
