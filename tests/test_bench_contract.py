"""bench.py's host-side pieces, exercised as code (no GPU): argument defaults, the name of the
selection kernel a launch runs, and the rule that a stored PMC traffic figure is only quoted
for the configuration it was measured on.  The bench line itself is checked on a GPU by
tests/test_bench_gpu.py, which runs bench.py."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("az_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_defaults_are_baseline_config_1(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert (a.gpus, a.games, a.n_playout, a.vl_batch, a.streams, a.evaluator, a.table) == (1, 8192, 200, 4, 1, "cnn", 0)
    assert a.steps > 0 and a.warmup >= 0 and not a.no_cpu_baseline
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "1"])
    a = b.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 5, 1)


def test_select_kernel_name_follows_the_variant(monkeypatch):
    b = _bench()
    monkeypatch.delenv("AZ_SELECT_VARIANT", raising=False)
    assert b.select_kernel_name() == "k_select8x4"
    monkeypatch.setenv("AZ_SELECT_VARIANT", "0")
    assert b.select_kernel_name() == "k_select<Connect4Dev,true>"
    monkeypatch.setenv("AZ_SELECT_VARIANT", "1")
    assert b.select_kernel_name() == "k_select8<true>"


def test_stored_traffic_is_keyed_by_configuration():
    doc = json.load(open(os.path.join(ROOT, "profiles", "traffic_select.json")))
    assert "by_config" in doc and "hbm_bytes_per_launch" not in doc, "a bare constant would be quoted for any configuration"
    for key, rec in doc["by_config"].items():
        for part in ("|games=", "|n_playout=", "|K=", "|streams=", "|evaluator=", "|lead_in="):
            assert part in key, key
        assert rec["hbm_bytes_per_launch"] == int((2 * rec["fetch_size_kib_raw"] + rec["write_size_kib"]) * 1024) or \
            abs(rec["hbm_bytes_per_launch"] - (2 * rec["fetch_size_kib_raw"] + rec["write_size_kib"]) * 1024) < 2048


def test_timed_region_of_bench_does_not_touch_the_oracle():
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert "oracle" not in main.replace("oracle/_ref", ""), "only the cpu_baseline leg may name the oracle"


def test_gpus_flag_disagreeing_with_the_launched_ranks_fails_loudly():
    """A rank whose WORLD_SIZE is not `--gpus` must not print a line for another job size (checked before torch is
    imported, so no GPU is needed to see it)."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()


def test_gpus_flag_starts_the_launcher_as_a_child(monkeypatch):
    """`--gpus N` without WORLD_SIZE: the parent builds a `torch.distributed.run --nproc-per-node N` command for
    itself, runs it as a CHILD (Popen, no exec), relays the line and insists on n_gpus == N."""
    import subprocess
    b = _bench()
    seen = {}

    class FakeChild:
        def __init__(self, cmd, **kw):
            seen["cmd"], seen["env"] = cmd, kw.get("env", {})
            self.stdout = iter(['{"n_gpus": %d, "value": 1.0}\n' % seen.get("report", 4)])

        def wait(self):
            return 0

    monkeypatch.setattr(subprocess, "Popen", FakeChild)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "1"])
    assert b.main() == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "5", "--warmup", "1"]
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0" or os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") is not None
    seen["report"] = 1                                # a run that reports another size is a failure
    import pytest
    with pytest.raises(SystemExit):
        b.main()
