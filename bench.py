#!/usr/bin/env python3
"""Headline benchmark: Connect4 self-play, n_playout=200, 8192 games per GPU, vl_batch=4
(BASELINE.json configs[1]) - positions/s and node-expansions/s on N GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With `--gpus N` (N > 1) and no launcher around it, bench.py starts the N ranks ITSELF: the parent - before
it imports torch or touches a GPU - runs `python -m torch.distributed.run --nproc-per-node N bench.py ...` as
a child process, relays rank 0's line and exits non-zero unless that line says `n_gpus: N`.  A rank whose
WORLD_SIZE differs from `--gpus` refuses to run.

A "step" is one ply played in every one of the 8192 games of a rank: 200 simulations per
tree (1 warm-up + 50 virtual-loss iterations of 4), each iteration = HIP selection + leaf
preparation -> the reference's CNN (random init; its hand-written HIP inference twin, bf16 with f32
accumulation as the reference's autocast: nn_*.hip, no PyTorch kernel) -> HIP
expansion/backup, then action sampling, re-rooting with fresh Dirichlet noise and the game
step, all resident in HBM (alphazero-al_amd/src/selfplay.py).  Ranks hold independent game
shards (weak scaling); the only collective is one all-reduce of the counters at the end.

Output: ONE JSON line on rank 0.  `value` = positions/s of the whole job.  `roofline` is for
the tree path's selection kernel - the path `north_star` asks an HBM roofline for; it is NOT the kernel
with the largest share of a step (the evaluator's kernels are: `roofline_evaluator_kernels` carries all of
them against both roofs): algorithmic bytes per launch (SURVEY.md
8(d): 28+29E per level with E=7, plus 16 B of leaf state per simulation, from the engine's
own level/simulation counters) over the kernel's mean duration measured with HIP events on
the launch stream during the timed region.  `cpu_baseline` times the reference's C++/OpenMP
search (oracle/_ref, compiled from the reference's sources) with the same network on the
host cores, on a bounded sample of the GPU leg's own positions; `tree_only` / `cpu_baseline_tree_only`
repeat both legs with the integer-hash evaluator (no network on either side).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "alphazero-al_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
SEL_BYTES_PER_LEVEL = 28 + 29 * 7   # parent 28 B + 7 x (edge 8 B + child 21 B)
LEAF_STATE_BYTES = 16               # two u64 bitboards written per simulation


T_START = time.perf_counter()


def log(msg):
    """Progress on stderr (the JSON line on stdout stays alone)."""
    sys.stderr.write("[bench %7.1fs] %s\n" % (time.perf_counter() - T_START, msg))
    sys.stderr.flush()


def host_cores():
    """CPU share of this process: affinity mask capped by the cgroup quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("AZ_CPU_CORES", "64"))))


PROFILE_EVERY = 16       # an event pair costs the stream ~5 us: six kinds of kernel are timed, so every 16th iteration carries them (0.3 % of a step)


def select_kernel_name(lib=None, handle=None):
    """The selection kernel behind the timed launches: asked of the engine (az_mcts_timed_select_kernel reports what
    launch_select actually launched); without an engine, what kernels.hip launch_select would pick for a
    virtual-loss batch of 2..4 descents under AZ_SELECT_VARIANT."""
    if lib is not None and handle is not None:
        import ctypes as C
        lib.az_mcts_timed_select_kernel.restype = C.c_char_p
        lib.az_mcts_timed_select_kernel.argtypes = [C.c_void_p]
        name = lib.az_mcts_timed_select_kernel(handle)
        if name:
            return name.decode()
    v = int(os.environ.get("AZ_SELECT_VARIANT", "3"))
    return {0: "k_select<Connect4Dev,true>", 1: "k_select8<true>", 2: "k_select8<true>"}.get(v, "k_select8x4")


def self_launch(args):
    """`--gpus N` with N > 1 and no launcher around this process: start the N ranks as a CHILD process
    (`python -m torch.distributed.run`, one rank per GPU) - never exec, and before this process has imported
    torch or touched a GPU - relay what the ranks print, and fail unless rank 0's line says n_gpus == N."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    log("--gpus %d without a launcher: starting %d ranks: %s" % (args.gpus, args.gpus, " ".join(cmd)))
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for out in child.stdout:
        # stdout of this process carries ONE line, the bench line; whatever else the ranks or their libraries print
        # on stdout (gloo's connection messages in a rehearsal) goes to stderr
        if out.startswith("{"):
            line = out
        else:
            sys.stderr.write(out)
            sys.stderr.flush()
    rc = child.wait()
    if rc != 0:
        raise SystemExit("bench.py: the %d-rank run exited with code %d" % (args.gpus, rc))
    try:
        seen = json.loads(line)["n_gpus"] if line else None
    except Exception:
        seen = None
    if seen != args.gpus:
        raise SystemExit("bench.py: asked for %d GPUs, the run reported n_gpus=%r" % (args.gpus, seen))
    sys.stdout.write(line)
    sys.stdout.flush()
    return 0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed plies (one ply in every game each); the default keeps the timed region at ~4 s")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--lead-in", type=int, default=12,
                    help="untimed plies played before the warm-up so that the timed region sees games of all ages, "
                         "as continuous self-play does (SURVEY 8d config C1: whole batches of games, not openings only); "
                         "0 = start the warm-up from empty boards")
    ap.add_argument("--games", type=int, default=8192, help="games (trees) per GPU")
    ap.add_argument("--n-playout", type=int, default=200)
    ap.add_argument("--vl-batch", type=int, default=4)
    ap.add_argument("--evaluator", choices=["cnn", "hash"], default="cnn",
                    help="cnn: the reference's network (headline); hash: integer hash evaluator (tree kernels only)")
    ap.add_argument("--table", type=int, default=0, metavar="LOG2",
                    help="device transposition table of 2^LOG2 evaluator outputs (the reference's cache_size, "
                         "BASELINE config 4); OFF for the headline number, which evaluates every leaf")
    ap.add_argument("--streams", type=int, default=1,
                    help="split the games of a GPU into this many independent drivers, each on its own HIP stream and "
                         "host thread (selfplay.StreamedSelfPlay): one group's tree kernels run under another group's "
                         "evaluator kernels; 1 = one driver, one batch of games x vl_batch leaves per iteration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tree-only", action="store_true", help="skip the tree-only pair (tree_only / cpu_baseline_tree_only)")
    ap.add_argument("--tree-only-steps", type=int, default=40)
    ap.add_argument("--cpu-games", type=int, default=256)
    ap.add_argument("--cpu-plies", type=int, default=6)
    return ap.parse_args()


def dump_positions(sp, n):
    """The first `n` games of the GPU leg as they stand (bitboards in HBM -> host arrays): the CPU leg plays on from
    the very positions the timed region ended on - games of all ages - instead of from empty boards."""
    import numpy as np
    part = sp.parts[0]
    n = min(int(n), part.B)
    return (part.bb_p1[:n].cpu().numpy().astype(np.uint64), part.bb_p2[:n].cpu().numpy().astype(np.uint64),
            part.turn[:n].cpu().numpy().astype(np.int32))


def cpu_baseline(args, evaluator, positions):
    """Reference C++/OpenMP search (oracle/_ref) + the same evaluator on the host CPU, through the
    reference-compatible wrapper, on a bounded sample of the same workload: `positions` are games of the GPU leg
    (dump_positions); a game that ends restarts from the empty board, as on the GPU."""
    import tempfile
    import numpy as np
    ref_dirs = [os.path.join(ROOT, "oracle", "_ref", v) for v in ("portable", "native")]
    ref_dir = next((d for d in ref_dirs if os.path.isdir(os.path.join(d, "src"))), None)
    if os.environ.get("AZ_BENCH_FORCE_PORT") == "1":
        ref_dir = None
    cores = host_cores()
    games, plies = args.cpu_games, args.cpu_plies
    if evaluator == "hash":            # tree-only leg: no network on the CPU either, so a sample of the default size lasts 40 ms
        games, plies = max(games, 4096), max(plies, 8)
    if ref_dir is None:
        # oracle/_ref did not travel: the plain-C restatement (oracle/, single-threaded search) takes its place
        kind, games = "port", min(games, 64)
        log(f"cpu_baseline[{evaluator}]: oracle C restatement (1 thread) + evaluator on {cores} host threads ...")
        head = f"""
import sys, time, json, os
import numpy as np, torch
sys.path[:0] = [{PKG!r}, {ROOT!r}]
from oracle import oracle as O                 # CPU restatement of the reference search (test infrastructure)
from src import MCTS_cpp as W
W._BACKENDS['Connect4'] = O.BatchedMCTS_Connect4
"""
    else:
        kind = "reference"
        log(f"cpu_baseline[{evaluator}]: reference C++/OpenMP search + evaluator on {cores} host threads ...")
        head = f"""
import sys, time, json, os
import numpy as np, torch
sys.path[:0] = [{ref_dir!r}, {PKG!r}]
from src import mcts_cpp                       # the REFERENCE's compiled module (oracle/_ref)
import importlib.util
spec = importlib.util.spec_from_file_location('az_wrap', os.path.join({PKG!r}, 'src', 'MCTS_cpp.py'))
W = importlib.util.module_from_spec(spec); spec.loader.exec_module(W)
"""
    tmp = None
    if positions is not None:
        bb1, bb2, turn = (np.asarray(a)[:games] for a in positions)
        games = int(bb1.shape[0])
        tmp = tempfile.NamedTemporaryFile(prefix="az_bench_positions_", suffix=".npz", delete=False)
        np.savez(tmp, bb1=bb1, bb2=bb2, turn=turn)
        tmp.close()
    code = head + f"""
import importlib.util
spec = importlib.util.spec_from_file_location('az_net', os.path.join({PKG!r}, 'src', 'az_net.py'))
N = importlib.util.module_from_spec(spec); spec.loader.exec_module(N)
torch.manual_seed(0); torch.set_num_threads({cores})
if {evaluator!r} == 'hash':
    spec = importlib.util.spec_from_file_location('az_hash', os.path.join({PKG!r}, 'src', 'hash_eval.py'))
    H = importlib.util.module_from_spec(spec); spec.loader.exec_module(H)
    net = H.NumpyHashEvaluator()                # tree-only leg: the integer-hash evaluator in numpy
else:
    net = N.Connect4Net(device='cpu').eval()
B, n, K, plies = {games}, {args.n_playout}, {args.vl_batch}, {plies}
w = W.BatchedMCTS(B, c_init=1.4, c_base=5*n, alpha=0.3, n_playout=n, noise_epsilon=0.25,
                  fpu_reduction=0.2, use_symmetry=True, mlh_slope=0.1, mlh_cap=0.2)
w.seed(0)
boards = np.zeros((B, 6, 7), np.int8); turns = np.ones(B, np.int32)
src = {(tmp.name if tmp else None)!r}
if src:
    z = np.load(src)
    for c in range(7):                          # bit 7*col + height; row 0 is the top of the board (Connect4.h:15-29)
        for h in range(6):
            bit = np.uint64(7 * c + h)
            boards[:, 5 - h, c] = ((z['bb1'] >> bit) & np.uint64(1)).astype(np.int8) - ((z['bb2'] >> bit) & np.uint64(1)).astype(np.int8)
    turns = z['turn'].astype(np.int32)
stones = int(np.abs(boards).sum())
def ended(g, r, c):                             # four in a row through (r, c), or a full board (Connect4.h:159-203)
    p = g[r, c]
    for dr, dc in ((0, 1), (1, 0), (1, 1), (1, -1)):
        run = 1
        for sg in (1, -1):
            rr, cc = r + sg * dr, c + sg * dc
            while 0 <= rr < 6 and 0 <= cc < 7 and g[rr, cc] == p:
                run += 1; rr += sg * dr; cc += sg * dc
        if run >= 4:
            return True
    return not (g == 0).any()
finished = 0
t0 = time.perf_counter()
for ply in range(plies):
    w.batch_playout(net, boards, turns, vl_batch=K, fused=False)
    c = w.get_visits_count(); a = c.argmax(1).astype(np.int32)
    w.prune_roots(a)
    restart = []
    for i in range(B):
        col = boards[i][:, a[i]]
        free = np.where(col == 0)[0]
        if free.size == 0:                      # cannot happen for a searched, unfinished position
            restart.append(i); continue
        r = free.max(); boards[i][r, a[i]] = turns[i]; turns[i] = -turns[i]
        if ended(boards[i], r, a[i]):
            restart.append(i)
    if restart:                                 # finished games restart from the empty board, their trees reset
        finished += len(restart)
        boards[restart] = 0; turns[restart] = 1
        for i in restart:
            w.reset_env(int(i))
dt = time.perf_counter() - t0
print(json.dumps(dict(value=B*plies/dt, seconds=dt, positions=B*plies, stones=stones, finished=finished)))
"""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS=str(cores), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        r = json.loads(line)
    except Exception as e:                      # the baseline is a reported extra, never fatal
        sys.stderr.write(f"cpu_baseline failed: {e}\n" + (out.stderr[-2000:] if "out" in dir() else ""))
        return None
    finally:
        if tmp is not None:
            try:
                os.unlink(tmp.name)
            except OSError:
                pass
    what = (f"reference C++/OpenMP search (oracle/_ref/{os.path.basename(ref_dir)})" if kind == "reference"
            else "oracle/ C restatement of the reference search, one thread")
    ev = (f"same CNN fp32 on CPU ({cores} torch threads)" if evaluator == "cnn"
          else "integer-hash evaluator in numpy (tree-only leg)")
    start = (f"continued from {games} of the GPU leg's own games as they stood after its timed region "
             f"(all ages: {r['stones'] / max(games, 1):.1f} stones per board on average; {r['finished']} games ended and restarted)"
             if positions is not None else f"{games} games from EMPTY boards")
    return {"value": round(r["value"], 2), "unit": "positions/s", "cores": cores, "kind": kind,
            "sample": f"{plies} plies, {start}, n_playout={args.n_playout}, vl_batch={args.vl_batch}, "
                      f"{what} + {ev}, {r['seconds']:.1f} s"}


def conv_roofline(torch, fast, leaves, launches=20):
    """The kernel with the largest share of a step (one residual convolution block of the
    evaluator, nn_conv.hip) against the matrix-core roofline: 2 * 42 * 64 * 576 flops per leaf,
    timed by events on the stream it is launched on, at the bench's leaf count."""
    import ctypes as C
    from src.fast_net import glue
    L = glue()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = (torch.randn((leaves, 42, 64), device=fast.device) * 0.5).to(torch.bfloat16)
    y = torch.empty_like(x)
    w, b, g, beta = (getattr(fast, n) for n in fast.res[0])

    def run():
        L.az_nn_conv_block(x.data_ptr(), 64, w.data_ptr(), b.data_ptr(), g.data_ptr(), beta.data_ptr(), 1, y.data_ptr(),
                           leaves, 1e-5, None, s)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / launches * 1e3
    flops = 2.0 * 42 * 64 * 576 * leaves
    achieved = flops / (us * 1e-6) / 1e12
    return {"bound": "mfma", "kernel": "k_conv_block<64,norm,residual> (3 of the 6 evaluator launches, ~45 % of a step)",
            "achieved": round(achieved, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(achieved / 2500.0, 4),
            "avg_launch_us": round(us, 1), "flops_per_launch": int(flops),
            "traffic": 2 * leaves * 42 * 64 * 2, "note": "dense bf16 MFMA peak; the kernel's HBM traffic equals its "
            "algorithmic bytes (profiles/r02_pmc_fetch_write_by_kernel.csv)"}


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args)                 # the parent never imports torch.cuda: its children own the GPUs
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d ranks were launched (rank %d): refusing to print a line "
                         "for another job size" % (args.gpus, world, rank))
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the search engine has no CPU path)")
    # rehearsal of the multi-rank path on a one-GPU box: AZ_BENCH_REHEARSE=1 puts every rank on
    # cuda:0 and exchanges the counters over gloo (RCCL cannot open one device twice)
    rehearse = os.environ.get("AZ_BENCH_REHEARSE", "0") == "1"
    if rehearse:
        local = 0
    elif local >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d wants cuda:%d, this node shows %d GPUs (one rank per GPU; "
                         "AZ_BENCH_REHEARSE=1 rehearses the multi-rank path on one GPU)" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))   # RCCL

    import __graft_entry__ as ge
    if rank == 0 and not os.path.exists(os.path.join(PKG, "lib", "libaz_mcts.so")):
        ge.build()
    if world > 1:
        dist.barrier()
    from src import fused as F
    from src.az_net import Connect4Net
    from src.selfplay import StreamedSelfPlay

    dev = torch.device("cuda", local)
    torch.manual_seed(1234)                     # same random-init weights on every rank
    if args.evaluator == "hash":
        from src.hash_eval import HashEvaluator
        net = HashEvaluator(dev)
    else:
        net = Connect4Net(device=dev).eval()
    sp = StreamedSelfPlay(net, args.games, streams=args.streams, n_playout=args.n_playout, vl_batch=args.vl_batch,
                          seed=rank, reserve_slots=int(os.environ["AZ_RESERVE_SLOTS"]) if "AZ_RESERVE_SLOTS" in os.environ else None,
                          table_log2=args.table)
    handles = [part.h for part in sp.parts]
    L = F.lib()

    log(f"rank {rank}: engine + evaluator ready ({args.games} games, n_playout={args.n_playout}, K={args.vl_batch})")
    sp.step(args.lead_in)
    if args.lead_in:
        torch.cuda.synchronize()
        log(f"lead-in: {args.lead_in} plies, {sp.read_totals()['games']} games finished and restarted")
    for i in range(args.warmup):
        tw = time.perf_counter()
        sp.step()
        torch.cuda.synchronize()
        log(f"warm-up step {i + 1}/{args.warmup}: {time.perf_counter() - tw:.2f} s")
    torch.cuda.synchronize()
    for h in handles:
        F.check(L.az_mcts_counters_reset(h))
        F.check(L.az_mcts_profile(h, PROFILE_EVERY))      # every 16th selection / backup launch carries an event pair
    import ctypes as C
    from src.fast_net import glue
    G = glue()
    G.az_nn_model_profile.argtypes = [C.c_int]
    G.az_nn_model_profile_read.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    G.az_nn_model_profile(PROFILE_EVERY)                   # and every 16th forward call of the evaluator, kernel by kernel
    t_before = sp.read_totals()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sp.step(args.steps)          # the drivers' host threads are joined once, after the last ply is enqueued
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    log(f"timed region: {args.steps} steps in {elapsed:.2f} s")

    G.az_nn_model_profile_read_kernels.argtypes = [C.POINTER(C.c_double * 4), C.POINTER(C.c_int64 * 4)]
    ev_ms, ev_n = (C.c_double * 4)(), (C.c_int64 * 4)()
    G.az_nn_model_profile_read_kernels(C.byref(ev_ms), C.byref(ev_n))
    G.az_nn_model_profile(0)
    conv_ms, conv_n = C.c_double(ev_ms[1]), C.c_int64(ev_n[1])
    ms = [0.0, 0.0]; nl = [0, 0]
    for h in handles:
        ms_h = (C.c_double * 2)(); nl_h = (C.c_int64 * 2)()
        F.check(L.az_mcts_profile_read(h, C.byref(ms_h), C.byref(nl_h)))
        F.check(L.az_mcts_profile(h, 0))
        for i in range(2):
            ms[i] += float(ms_h[i]); nl[i] += int(nl_h[i])
    cnt = sp.engine_counters()
    tot = sp.read_totals()
    positions = tot["positions"] - t_before["positions"]
    games = tot["games"] - t_before["games"]

    # one collective: sum the counters, max the time (xGMI, a few dozen bytes)
    from src.shard import reduce_counters
    totals, t = reduce_counters([positions, cnt["sims"], cnt["expansions"], games, cnt["levels"],
                                 cnt["backup_nodes"]], elapsed, torch.device("cpu") if rehearse else dev)
    g_pos, g_sims, g_exp, g_games = totals["positions"], totals["sims"], totals["expansions"], totals["games"]

    if rank == 0:
        sims_rank = max(cnt["sims"], 1)
        depth = cnt["levels"] / sims_rank
        xps = cnt["expansions"] / sims_rank
        sel_ms, sel_n = float(ms[0]), int(nl[0])
        bp_ms, bp_n = float(ms[1]), int(nl[1])
        sel_bytes = cnt["levels"] * SEL_BYTES_PER_LEVEL + cnt["sims"] * LEAF_STATE_BYTES
        roofline = None
        pair_ms = 0.0
        if sel_n > 0 and sel_ms > 0:
            # counters cover every launch of the timed region; events every PROFILE_EVERY-th of them
            launches = max(cnt["select_launches"], 1)
            per_launch_bytes = sel_bytes / launches
            # An event pair brackets more than the kernel: the two records take time on the stream
            # themselves.  That constant is measured here with empty pairs and taken off, which is
            # what makes the figure agree with rocprofv3's kernel durations (profiles/README.md).
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(128)]
            for i in range(0, 128, 2):
                ev[i].record(); ev[i + 1].record()
            torch.cuda.synchronize()
            gaps = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(0, 128, 2))
            pair_ms = gaps[len(gaps) // 2]
            raw_ms = sel_ms / sel_n
            avg_ms = max(raw_ms - pair_ms, raw_ms * 0.25)
            achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9
            # HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes,
            # tools/collect_traffic.py): a stored measurement is only quoted for the configuration it was
            # taken on - kernel, trees per launch, n_playout, K, streams, evaluator - otherwise null
            traffic = None
            sel_kernel = select_kernel_name(L, handles[0])
            traffic_key = "%s|games=%d|n_playout=%d|K=%d|streams=%d|evaluator=%s|lead_in=%d" % (
                sel_kernel, args.games, args.n_playout, args.vl_batch, args.streams, args.evaluator, args.lead_in)
            tf = os.path.join(ROOT, "profiles", "traffic_select.json")
            if os.path.exists(tf):
                try:
                    traffic = json.load(open(tf)).get("by_config", {}).get(traffic_key, {}).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            # achievable copy bandwidth on this box (SURVEY 8d asks for it beside the datasheet peak)
            src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
            dst = torch.empty_like(src)
            dst.copy_(src)
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            for _ in range(5):
                dst.copy_(src)
            c1.record()
            torch.cuda.synchronize()
            copy_gbs = 5 * 2 * (1 << 30) / (c0.elapsed_time(c1) * 1e-3) / 1e9
            del src, dst
            roofline = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_key": traffic_key,
                        "traffic_source": "profiles/traffic_select.json: PMC passes (FETCH_SIZE x 2 + WRITE_SIZE) taken on exactly this "
                                          "configuration by tools/collect_traffic.py; null for any other" if traffic is not None else None,
                        "kernel": sel_kernel, "avg_launch_us": round(avg_ms * 1e3, 2),
                        "bound_note": "priced against the HBM roof as north_star asks for the tree walk; the counters say the kernel is "
                                      "limited by vector-instruction issue (DESIGN.md section 3), and it is a few percent of a step - the "
                                      "kernels that dominate a step are in roofline_evaluator_kernels",
                        "avg_event_pair_us": round(raw_ms * 1e3, 2), "empty_event_pair_us": round(pair_ms * 1e3, 2),
                        "launches_timed": sel_n, "algorithmic_bytes_per_launch": int(per_launch_bytes),
                        "measured_copy_GBs": round(copy_gbs, 1), "frac_of_measured_copy": round(achieved / copy_gbs, 6),
                        "trees_per_launch": args.games // max(args.streams, 1), "concurrent_streams": args.streams}
            if args.streams > 1:
                roofline["note"] = ("%d drivers on separate streams: a launch covers %d trees and shares the chip with the "
                                    "other drivers' evaluator kernels while it runs, so its duration is not the kernel's "
                                    "isolated time (--streams 1: one launch over all trees, profiles/README.md)"
                                    % (args.streams, args.games // args.streams))
        out = {
            "metric": "self-play positions/sec (Connect4 n_playout=%d, batch=%d games/GPU, vl_batch=%d%s)"
                      % (args.n_playout, args.games, args.vl_batch, "" if args.evaluator == "cnn" else ", tree kernels only: integer-hash evaluator"),
            "value": round(g_pos / t, 2), "unit": "positions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(t / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "dtype_detail": "tree statistics f32 (positions u64 bitboards, counts i32); network bf16 with f32 accumulation, as the reference's autocast",
            "data": ("synthetic: continuous self-play (games of all ages after a %d-ply lead-in from empty boards), "
                     % args.lead_in if args.lead_in else "synthetic: self-play from empty boards, ")
                    + ("random-init network" if args.evaluator == "cnn" else "integer-hash evaluator"),
            "config": {"workload": "Connect4 self-play, n_playout=%d, %d games/GPU, vl_batch=%d, evaluator=%s"
                                   % (args.n_playout, args.games, args.vl_batch, args.evaluator),
                       "c_init": 1.4, "c_base": 5 * args.n_playout, "fpu_reduction": 0.2, "dirichlet_alpha": 0.3,
                       "noise_epsilon": 0.25, "mlh_slope": 0.1, "use_symmetry": True, "lead_in_plies": args.lead_in,
                       "streams_per_gpu": args.streams,
                       "parallelism": "independent game shards x%d" % world},
            "sims_per_s": round(g_sims / t, 1), "node_expansions_per_s": round(g_exp / t, 1),
            "node_expansions_per_s_per_gpu": round(g_exp / t / world, 1),
            "games_finished": g_games, "mean_select_depth": round(depth, 3), "expansions_per_sim": round(xps, 3),
            # (event pairs with the empty pair's cost taken off, as for the selection kernel: agrees with rocprofv3's durations)
            "backprop_kernel_avg_us": round(max(bp_ms / bp_n - pair_ms, bp_ms / bp_n * 0.25) * 1e3, 2) if bp_n else None,
            "backprop_event_pair_us": round(bp_ms / bp_n * 1e3, 2) if bp_n else None,
            "tree_kernels_share_of_step": round(((max(sel_ms / sel_n - pair_ms, 0.0) if sel_n else 0.0) * cnt["select_launches"] +
                                                 (max(bp_ms / bp_n - pair_ms, 0.0) if bp_n else 0.0) * cnt["backprop_launches"])
                                                / (elapsed * 1e3) / max(args.streams, 1), 4),
            "roofline": roofline,
        }
        fast = sp.parts[0].fused.fast
        if args.evaluator == "cnn" and fast is not None and getattr(fast, "mfma_conv", False):
            synth = conv_roofline(torch, fast, args.games * args.vl_batch // max(args.streams, 1))
            if conv_n.value > 0:
                # measured on launches of the timed region: the evaluator sees the non-terminal leaves only
                live = (cnt["sims"] - cnt["terminal"]) / max(cnt["select_launches"], 1)
                raw_us = conv_ms.value / conv_n.value * 1e3
                us = max(raw_us - (roofline["empty_event_pair_us"] if roofline else 0.0), raw_us * 0.25)
                flops = 2.0 * 42 * 64 * 576 * live
                ach = flops / (us * 1e-6) / 1e12
                out["roofline_evaluator"] = {
                    "bound": "mfma", "kernel": "k_conv_block<64,norm,residual> (3 of the 6 evaluator launches, ~43 % of a step)",
                    "achieved": round(ach, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(ach / 2500.0, 4),
                    "avg_launch_us": round(us, 1), "launches_timed": int(conv_n.value), "leaves_per_launch": round(live, 1),
                    "flops_per_launch": int(flops), "traffic": int(2 * live * 42 * 64 * 2),
                    "note": "HIP events around every %d-th first residual block inside the timed region (az_nn_model_profile), "
                            "event-pair overhead removed; dense bf16 MFMA peak; the kernel's HBM traffic equals its algorithmic "
                            "bytes (profiles/r02_pmc_fetch_write_by_kernel.csv)" % PROFILE_EVERY,
                    "synthetic_launch": {"achieved": synth["achieved"], "avg_launch_us": synth["avg_launch_us"],
                                         "leaves": args.games * args.vl_batch // max(args.streams, 1),
                                         "note": "same kernel on random activations after the run: slower, the clock follows the data"}}
                # every kernel of the evaluator, same event pairs, same timed region.  FLOPs per leaf from the layer
                # shapes (Network.py:27-93,96-141): stem 42 x 64 x 288 MACs, residual block 42 x 64 x 576, attention
                # 42 x 196 x 64 (QKVG) + 2 x 4 x 42 x 42 x 16 (scores, PV) + 42 x 64 x 64 (out); the heads are
                # pooling + 64-wide linears on a few vectors: memory traffic, not matrix work, bounds them.
                act = 42 * 64 * 2                                  # one sample's bf16 activations
                pair = roofline["empty_event_pair_us"] if roofline else 0.0
                table = []
                for idx, name, flops_leaf, bytes_leaf in (
                        # the stem is a K = 18 (one k step of 32, table in a bf16 high and low part) GEMM on a 0/1 operand
                        # built from two bitboards (nn_stem.hip): 24 + 8 bytes in, one sample's activations out
                        ((0, "k_stem (embedding + stem convolution from the bitboards)", 2.0 * 2 * 48 * 64 * 32, 32 + act)
                         if getattr(fast, "folded_stem", False) else
                         (0, "k_conv_block<32,..,EMBED> (stem + embedding)", 2.0 * 42 * 64 * 288, 504 + act)),
                        (1, "k_conv_block<64,norm,residual> (x3 per forward)", 2.0 * 42 * 64 * 576, 2 * act),
                        (2, "k_attn_block", 2.0 * (42 * 196 * 64 + 2 * 4 * 42 * 42 * 16 + 42 * 64 * 64), 2 * act),
                        (3, "k_heads", 2.0 * (42 * 64 + 9 * 64 * 64 + 4 * 64 * 64 + 64 * 46), act + 7 + 44)):
                    if ev_n[idx] <= 0:
                        continue
                    raw = ev_ms[idx] / ev_n[idx] * 1e3
                    us_k = max(raw - pair, raw * 0.25)
                    tf_s = flops_leaf * live / (us_k * 1e-6) / 1e12
                    gb_s = bytes_leaf * live / (us_k * 1e-6) / 1e9
                    bound = "mfma" if tf_s / 2500.0 >= gb_s / HBM_PEAK_GBS else "hbm"
                    table.append({"kernel": name, "avg_launch_us": round(us_k, 1), "launches_timed": int(ev_n[idx]),
                                  "bound": bound,
                                  "achieved": round(tf_s if bound == "mfma" else gb_s, 1),
                                  "peak": 2500.0 if bound == "mfma" else HBM_PEAK_GBS,
                                  "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                                  "frac": round(max(tf_s / 2500.0, gb_s / HBM_PEAK_GBS), 4),
                                  "frac_mfma": round(tf_s / 2500.0, 4), "frac_hbm": round(gb_s / HBM_PEAK_GBS, 4),
                                  "flops_per_launch": int(flops_leaf * live), "algorithmic_bytes_per_launch": int(bytes_leaf * live)})
                out["roofline_evaluator_kernels"] = table
            else:
                out["roofline_evaluator"] = synth
        if args.table:
            st = sp.table_stats()                        # whole run, warm-up included
            out["config"]["workload"] += ", transposition table 2^%d entries" % args.table
            out["transposition_table"] = {"entries": 1 << args.table, "lookups": st["lookups"], "hits": st["hits"],
                                          "hit_rate": round(st["hit_rate"], 4), "replaced": st["replaced"]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.evaluator, dump_positions(sp, args.cpu_games))
            if args.evaluator == "cnn" and not args.no_tree_only:
                # the same pair with the network factored out: tree kernels only (integer-hash evaluator, native on
                # the GPU, numpy on the CPU) - GPU search against the reference's search, nothing else in the way
                sp.close()
                del sp, handles
                torch.cuda.empty_cache()
                from src.hash_eval import HashEvaluator
                sp2 = StreamedSelfPlay(HashEvaluator(dev), args.games, streams=1, n_playout=args.n_playout,
                                       vl_batch=args.vl_batch, seed=rank)
                sp2.step(args.lead_in + 1)
                torch.cuda.synchronize()
                before = sp2.read_totals()["positions"]
                F.check(L.az_mcts_counters_reset(sp2.parts[0].h))
                t1 = time.perf_counter()
                sp2.step(args.tree_only_steps)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                c2 = sp2.engine_counters()
                pos2 = sp2.read_totals()["positions"] - before
                out["tree_only"] = {"value": round(pos2 / dt, 1), "unit": "positions/s", "steps": args.tree_only_steps,
                                    "ms_per_step": round(dt / args.tree_only_steps * 1e3, 3),
                                    "sims_per_s": round(c2["sims"] / dt, 1), "node_expansions_per_s": round(c2["expansions"] / dt, 1),
                                    "evaluator": "integer-hash evaluator (native, no network): selection + leaf preparation + backup + game logic"}
                out["cpu_baseline_tree_only"] = cpu_baseline(args, "hash", dump_positions(sp2, max(args.cpu_games, 4096)))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
