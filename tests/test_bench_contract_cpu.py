"""The bench line contract (driver prompt + tier addendum) checked on the last bench line recorded on the MI355X
(profiles/round1_bench_call38.json): bench.py cannot run without a GPU, so the CPU suite pins the shape of what it printed."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
    files = glob.glob(os.path.join(ROOT, "profiles", "round*_bench_call*.json"))
    assert files
    def key(p):   # round, then the call numbering (round 4 restarted it: ..._call_r4_NN follows ..._callNNN)
        m = re.search(r"round(\d+)_bench_call(_r\d+_)?(\d+)", os.path.basename(p))
        return (int(m.group(1)), 1 if m.group(2) else 0, int(m.group(3)))
    return max(files, key=key)


def test_recorded_bench_line_has_the_contract_fields():
    d = json.load(open(_latest()))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port")
    # value = whole-job throughput: descriptor pairs of the workload / time per step
    m = re.search(r"\((\d+) image pairs", d["config"]["workload"])
    pairs = int(m.group(1))
    assert abs(d["value"] - pairs * 2000 * 2000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


def test_bench_sources_parse_and_default_to_one_gpu():
    import ast
    for name in ("bench.py", "bench_ba.py", "bench_hamming.py", "__graft_entry__.py"):
        ast.parse(open(os.path.join(ROOT, name)).read())
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"--gpus", type=int, default=1' in src and '"--steps"' in src and '"--warmup"' in src
