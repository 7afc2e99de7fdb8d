"""The N>1 path on CPU: two processes, gloo, 127.0.0.1 - shard layout and the single
end-of-run reduction that bench.py performs over RCCL on GPUs."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "alphazero-al_amd")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


WORKER = textwrap.dedent("""
    import importlib.util, json, os, sys
    import torch.distributed as dist
    spec = importlib.util.spec_from_file_location("shard", os.path.join(sys.argv[1], "src", "shard.py"))
    shard = importlib.util.module_from_spec(spec); spec.loader.exec_module(shard)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    first, n, seed = shard.shard_of(rank, world, games_per_rank=8192)
    f2, n2, _ = shard.shard_of(rank, world, total_games=16385)
    dist.barrier()
    counters = [n, n * 200, n * 190 + rank, 10 + rank, n * 600, n * 800]
    tot, tmax = shard.reduce_counters(counters, 1.0 + rank)
    dist.barrier()
    print(json.dumps(dict(rank=rank, first=first, n=n, seed=seed, f2=f2, n2=n2, tot=tot, tmax=tmax)))
    dist.destroy_process_group()
""")


def test_two_rank_gloo_reduction(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), PKG], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err
        outs.append(__import__("json").loads(out.strip().splitlines()[-1]))
    outs.sort(key=lambda o: o["rank"])
    # weak scaling: same games per rank, disjoint ranges, per-rank seed
    assert [(o["first"], o["n"], o["seed"]) for o in outs] == [(0, 8192, 0), (8192, 8192, 1)]
    # strong scaling split covers every game exactly once
    assert [(o["f2"], o["n2"]) for o in outs] == [(0, 8193), (8193, 8192)]
    for o in outs:       # every rank sees the same totals
        assert o["tot"]["positions"] == 16384 and o["tot"]["sims"] == 16384 * 200
        assert o["tot"]["expansions"] == 8192 * 190 * 2 + 1 and o["tot"]["games"] == 21
        assert o["tmax"] == 2.0
