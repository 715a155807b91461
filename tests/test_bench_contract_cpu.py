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


def test_the_driver_line_of_every_stored_full_record_is_small_and_complete():
    """the round-4 line was one 22 KB object and the driver did not recover it. bench.py now ends stdout with
    bench_line.compact(full record): < 4 KB, the headline fields of the contract, the BA figures and a digest per side record."""
    import bench_line
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round[45]_bench_call*.json")))
    assert files
    checked = 0
    for p in files:
        full = json.load(open(p))
        if "metric" not in full or "roofline" not in full:
            continue
        line, s = bench_line.compact(full, "gpurun_out/bench_side.json")
        assert len(s) < bench_line.MAX_LINE_BYTES and "\n" not in s, (p, len(s))
        d = json.loads(s)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                  "data", "config", "roofline"):
            assert k in d, (p, k)
        assert "workload" in d["config"] and abs(d["value"] - full["value"]) / full["value"] < 1e-8
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "mean_launch_ms"):
            assert k in d["roofline"], (p, k)
        if "cpu_baseline" in full:
            for k in ("value", "unit", "cores", "kind", "sample"):
                assert k in d["cpu_baseline"], (p, k)
            assert d["parity"]["identical"] == full["parity"]["identical"]
        for k in ("ba", "ba_c5_single_gpu"):
            if k in full and "lm_iteration_ms" in full[k]:
                b = d[k]
                assert b["final_rmse"] == full[k]["final_rmse"] and "frac" in b["roofline"] and "traffic_over_algorithmic" in b["roofline"]
                if "cpu_baseline" in full[k] and "value" in full[k]["cpu_baseline"]:
                    assert b["cpu_baseline"]["rmse_diff_vs_reference"] is not None and b["cpu_baseline"]["kind"] == "reference"
        checked += 1
    assert checked >= 1


def test_a_record_too_large_for_the_line_keeps_the_headline():
    import bench_line
    full = json.load(open(_latest()))
    full["hamming"] = dict(full.get("hamming") or {}, value=1.0)
    full["scale_selfcheck"] = {"ok": True, "blob": "x" * 650, "matching": {"m": "y" * 5000}}
    line, s = bench_line.compact(full, None)
    assert len(s) < bench_line.MAX_LINE_BYTES and line["value"] and line["roofline"]["frac"]


def test_bench_ends_stdout_with_the_compact_line():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "print(json.dumps(out)" not in src and src.count("emit(out, args)") == 3 and "bench_line.compact" in src


def test_bench_sources_parse_and_default_to_one_gpu():
    import ast
    for name in ("bench.py", "bench_ba.py", "bench_hamming.py", "bench_geofilter.py", "bench_line.py", "__graft_entry__.py"):
        ast.parse(open(os.path.join(ROOT, name)).read())
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"--gpus", type=int, default=1' in src and '"--steps"' in src and '"--warmup"' in src
