"""Multi-GPU layout of self-play: independent game shards, one process per GPU.

Games (one tree + one position each) never interact (the reference runs them under an
`omp parallel for`, BatchedMCTS.h:107-332), so ranks share nothing on the data path.  The only
collective of a run is the reduction of a handful of counters at the end: SUM of the int64
work counters and MAX of the elapsed time, over RCCL/xGMI on GPUs (backend "nccl") or gloo in
the CPU tests.
"""
import torch
import torch.distributed as dist

COUNTER_NAMES = ("positions", "sims", "expansions", "games", "levels", "backup_nodes")


def shard_of(rank, world, total_games=None, games_per_rank=None):
    """(first_game, n_games, seed) of a rank.  Weak scaling (`games_per_rank`) gives every rank
    the same number of games; strong scaling splits `total_games` as evenly as possible."""
    if games_per_rank is not None:
        return rank * games_per_rank, games_per_rank, rank
    base, extra = divmod(total_games, world)
    n = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, n, rank


def reduce_counters(counters, elapsed_s, device="cpu"):
    """All ranks pass their own counters (sequence of ints in COUNTER_NAMES order) and elapsed
    seconds; every rank gets (summed counters as a dict, max elapsed)."""
    vec = torch.tensor(list(counters), dtype=torch.int64, device=device)
    tmax = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    return dict(zip(COUNTER_NAMES, (int(v) for v in vec.tolist()))), float(tmax.item())
