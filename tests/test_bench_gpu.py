"""bench.py run for real on the GPU box: the one-rank line carries the contract's fields with
consistent numbers, and the multi-rank path (one process per shard, counters reduced once at the
end) is rehearsed with two ranks on one GPU over gloo - RCCL cannot open one device twice; the
reduction code is the same call with another backend (src/shard.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


SMALL = ["--games", "1024", "--steps", "3", "--warmup", "1", "--lead-in", "4"]


def test_one_rank_line_has_the_contract_fields():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *SMALL, "--cpu-games", "8", "--cpu-plies", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["unit"] == "positions/s" and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["vs_baseline"] is None
    assert "workload" in j["config"] and "model" not in j["config"] and "batch=1024" in j["metric"]
    assert abs(j["value"] - 1024 * 3 / (j["ms_per_step"] * 3 * 1e-3)) / j["value"] < 1e-3
    assert j["sims_per_s"] == pytest.approx(j["value"] * 200, rel=1e-3)
    r_ = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_key"):
        assert k in r_, k
    assert r_["bound"] == "hbm" and r_["unit"] == "GB/s" and abs(r_["frac"] - r_["achieved"] / r_["peak"]) < 1e-5
    assert r_["traffic"] is None, "no PMC measurement exists for this small configuration: nothing may be quoted"
    assert abs(r_["achieved"] - r_["algorithmic_bytes_per_launch"] / (r_["avg_launch_us"] * 1e-6) / 1e9) / r_["achieved"] < 1e-2
    c = j["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["unit"] == j["unit"] and c["cores"] >= 1 and c["value"] > 0
    kinds = {e["kernel"].split("<")[0].split(" ")[0] for e in j["roofline_evaluator_kernels"]}
    assert {"k_conv_block", "k_attn_block", "k_heads"} <= kinds
    for e in j["roofline_evaluator_kernels"]:
        assert 0 < e["frac"] < 1 and e["bound"] in ("mfma", "hbm") and e["launches_timed"] > 0


def test_two_rank_rehearsal_reduces_the_counters():
    """`torch.distributed.run` with two ranks, both on cuda:0 (AZ_BENCH_REHEARSE=1, gloo): each rank plays its own
    1024 games; the line is the whole job's."""
    env = dict(os.environ, AZ_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL, "--evaluator", "hash"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak"
    positions = j["value"] * j["ms_per_step"] * 1e-3 * j["steps"]
    assert abs(positions - 2 * 1024 * 3) < 2 * 1024 * 3 * 1e-3       # both shards' plies over the max of the two clocks (rounded fields)
    assert j["sims_per_s"] == pytest.approx(j["value"] * 200, rel=1e-3)
    assert j["node_expansions_per_s_per_gpu"] == pytest.approx(j["node_expansions_per_s"] / 2, rel=1e-6)
    assert "cpu_baseline" not in j                                  # rank 0 at N = 1 only


def test_gpus_flag_launches_the_ranks_itself():
    """`python bench.py --gpus 2` with NO launcher on the command line: the parent starts torch.distributed.run as a
    child process, both ranks play their shard (here on the one GPU of the box: AZ_BENCH_REHEARSE=1), and the
    relayed line is the two-rank job's."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["AZ_BENCH_REHEARSE"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL, "--evaluator", "hash"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak"
    positions = j["value"] * j["ms_per_step"] * 1e-3 * j["steps"]
    assert abs(positions - 2 * 1024 * 3) < 2 * 1024 * 3 * 1e-3
    assert "cpu_baseline" not in j


def test_rccl_backend_reduces_the_counters_single_rank():
    """The collective of a multi-GPU run - one all-reduce SUM of six int64 counters and one MAX of a float64,
    on device tensors over RCCL (`backend="nccl"`) - executed for real on this box's GPU with a one-rank
    process group: the same `shard.reduce_counters` call bench.py makes, the same backend, one device."""
    code = (
        "import os, sys, json, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "from src import shard\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group(backend='nccl', device_id=torch.device('cuda', 0))\n"
        "class Two:            # make reduce_counters take the collective branch with one rank\n"
        "    pass\n"
        "orig = dist.get_world_size\n"
        "dist.get_world_size = lambda *a, **k: 2 if not a and not k else orig(*a, **k)\n"
        "tot, t = shard.reduce_counters([8192, 1638400, 1343488, 77, 5300000, 6900000], 1.25, torch.device('cuda', 0))\n"
        "dist.get_world_size = orig\n"
        "dist.barrier()\n"
        "print(json.dumps(dict(tot=tot, t=t)))\n"
        "dist.destroy_process_group()\n" % os.path.join(ROOT, "alphazero-al_amd"))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _last_json(r.stdout)
    assert j["tot"]["positions"] == 8192 and j["tot"]["sims"] == 1638400 and j["tot"]["backup_nodes"] == 6900000
    assert j["t"] == 1.25
