"""bench.py --gpus 2 without the hardware: the driver's N > 1 launch line -
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ... - with
--rehearsal: gloo instead of RCCL, the HIP emulation libraries (the same device source compiled for the host) instead of libmvgx_hip.so,
the BA exchange through the library's callback transport over gloo, a tiny image set and BA scene. What runs is the control flow of
bench.py / bench_ba.py / tools/scale_selfcheck.py that no box has ever executed with N > 1: the gloo wait group around the self-check
subprocess, pair sharding, max-over-ranks timing, the per-rank parity turn-taking, the sharded BA leg beside the reference on the
full scene, the watchdog. The numbers of such a line mean nothing; its shape and its parity fields do."""
import json
import os
import socket
import subprocess
import sys

import pytest

from tests import _oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_bench(n, extra=()):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
           "--rehearsal", "--side-deadline", "600", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    # the driver's contract: stdout ENDS with the one line, and the line survives a bounded capture of the tail
    assert r.stdout.rstrip("\n").splitlines()[-1] == lines[0] and len(lines[0]) < 4096, len(lines[0])
    return json.loads(lines[0])


@pytest.mark.skipif(not os.path.isdir(os.path.join(ROOT, "tests", "native", "hipemu")), reason="HIP emulation sources absent")
def test_bench_two_ranks_rehearsal_prints_one_well_formed_line():
    d = _run_bench(2)
    assert d["rehearsal"] is True and d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    for k in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["scaling"] == "strong" and d["value"] > 0 and "pair-sharded x2" in d["config"]["parallelism"]
    # whole-job value: all pairs of the set / max-over-ranks time
    assert abs(d["value"] - 45 * 96 * 96 * 2 / (d["ms_per_step"] * 1e-3 * 2)) / d["value"] < 1e-6
    # every rank checked a sample of ITS shard against the CPU path
    par = d["parity"]
    assert par["ranks_checked"] == 2 and par["ranks_differing"] == 0 and par["identical"] is True and par["pairs_checked"] >= 2
    # the self-check subprocess ran (sharded in-process forms against one device) while the other rank waited in the gloo group
    sc = d["scale_selfcheck"]
    assert sc["ok"] is True and sc["matching"]["identical_to_one_device"] is True and sc["ba"]["peer"]["agrees"] is True
    # the BA leg: both ranks took part in the exchange, rank 0 ran the reference on the full scene beside the sharded solve
    ba = d["ba"]
    assert ba["exchange"] == {"ranks": 2, "transport": "callback over gloo (rehearsal)"} and ba["rccl_ranks"] == 0
    assert ba["iterations"] >= 1 and ba["final_rmse"] < ba["initial_rmse"]
    if _oracle.have_ref_ba():
        cb = ba["cpu_baseline"]
        assert cb["kind"] == "reference" and cb["rmse_diff_vs_reference"] < 1e-6, cb
