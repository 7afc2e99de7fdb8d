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
