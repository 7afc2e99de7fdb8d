"""The bench line's contract (task statement, section 4), checked on the line committed under
profiles/ (produced by `python bench.py` on an MI355X) and on bench.py's argument defaults - no GPU."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    txt = open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1]
    return json.loads(txt)


def test_committed_bench_line_has_the_contract_fields():
    j = _line("r01_bench_final.json")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["unit"] == "positions/s" and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["n_gpus"] == 1 and j["vs_baseline"] is None and "workload" in j["config"] and "model" not in j["config"]
    assert "positions" in str(base.get("metric", "")).lower() or "positions" in j["metric"]
    assert abs(j["value"] - 8192 * j["steps"] / (j["ms_per_step"] * j["steps"] * 1e-3)) / j["value"] < 1e-3
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) / r["achieved"] < 1e-3
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes_per_launch"]
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["unit"] == j["unit"] and c["cores"] >= 1
    e = j["roofline_evaluator"]
    assert e["bound"] == "mfma" and abs(e["frac"] - e["achieved"] / e["peak"]) < 1e-3


def test_bench_defaults_follow_the_contract():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert re.search(r'"--gpus", type=int, default=1', src)
    assert re.search(r'"--games", type=int, default=8192', src) and re.search(r'"--n-playout", type=int, default=200', src)
    assert re.search(r'"--vl-batch", type=int, default=4', src) and re.search(r'"--streams", type=int, default=1', src)
    assert "torch.cuda.synchronize()" in src and "dist.barrier()" in src
    # the product path never imports the oracle: only the cpu_baseline leg may name it
    main = src[src.index("def main():"):]
    assert "oracle" not in main.replace("oracle/_ref", "")
