"""BA leg of bench.py: BASELINE.json configs[2] — synthetic BA, 200 pinhole cameras, 100k points, ~1M observations,
LM iteration time + final RMSE on 1 MI355X, with the reference's Ceres CPU path timed beside it."""
import os
import time

import numpy as np

HBM_PEAK_GBS = 8000.0


def algorithmic_bytes_per_iteration(n_obs, n_pts, n_cam, n_cols):
    """fused fp64 LM iteration — observations read twice (Jacobian pass + candidate cost pass), points
    read + written, camera blocks, dense reduced system accumulated + factored, rhs."""
    return n_obs * (16 + 12) * 2 + n_pts * (24 + 24) + n_cam * 9 * 8 * 2 + 2 * (n_cols * n_cols * 8) + n_cols * 8


def _capture_stderr(fn):
    """Runs fn() with fd 2 redirected to a temp file (the reference logs Ceres' FullReport through its stderr logger)."""
    import tempfile
    with tempfile.TemporaryFile(mode="w+b") as tmp:
        saved = os.dup(2)
        try:
            os.dup2(tmp.fileno(), 2)
            out = fn()
        finally:
            os.dup2(saved, 2)
            os.close(saved)
        tmp.seek(0)
        return tmp.read().decode("utf-8", "replace"), out


def _parse_report(txt):
    """Seconds per phase from ceres::Solver::Summary::FullReport()."""
    import re
    out = {}
    for key, pat in (("preprocessor_s", r"Preprocessor\s+([0-9.]+)"), ("residual_eval_s", r"Residual evaluation\s+([0-9.]+)"),
                     ("jacobian_eval_s", r"Jacobian evaluation\s+([0-9.]+)"), ("linear_solver_s", r"Linear solver\s+([0-9.]+)\n"),
                     ("minimizer_s", r"Minimizer\s+([0-9.]+)\n"), ("total_s", r"Total\s+([0-9.]+)"),
                     ("iterations", r"Minimizer iterations\s+([0-9]+)"), ("threads", r"Threads\s+([0-9]+)")):
        m = re.search(pat, txt)
        if m:
            out[key] = float(m.group(1))
    return out


def ba_config(world, name=None):
    """BASELINE.json configs[2] at 1 GPU (200 pinhole cams / 100k points / 1M obs) growing to configs[4] at 8 GPUs
    (1k cams pinhole+K3 in 8 intrinsic groups / 500k points / 5M obs); linear in between. name="c5": configs[4] as is."""
    if name == "c5":
        return dict(n_cams=1000, n_points=500000, track_len=10, model=3, n_intr_groups=8, seed=0xBA5E0005)
    if world <= 1:
        return dict(n_cams=200, n_points=100000, track_len=10, model=1, n_intr_groups=1, seed=0xBA5E0003)
    f = min(1.0, (world - 1) / 7.0)
    return dict(n_cams=int(round(200 + 800 * f)), n_points=int(round(100000 + 400000 * f)), track_len=10, model=3,
                n_intr_groups=8, seed=0xBA5E0005)


def ba_bench_record(local_rank, world, cpu=True, cpu_budget_s=40.0, name=None):
    """One BA solve per rank on its point shard; returns the record on every rank (identical numbers: the LM state is
    replicated). world > 1 needs torch.distributed initialised (used only to hand out the RCCL unique id)."""
    from openmvg_amd import ba, sharding, synth
    cfg = ba_config(world, name)
    full = synth.ba_scene(**cfg)
    rank = 0
    if world > 1:
        import torch.distributed as dist
        rank = dist.get_rank()
        scene, _mine = sharding.shard_ba_scene(full, rank, world)
    else:
        scene = full

    def make():
        c = ba.BaContext(scene, device=local_rank)
        if world > 1:   # a unique id creates exactly one communicator: every context gets its own
            box = [ba.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            c.comm_init(world, rank, box[0])
        return c

    t0 = time.perf_counter()
    ctx = make()
    create_s = time.perf_counter() - t0
    s1 = ctx.lm_iteration(ba.default_options(max_num_iterations=1))   # iteration zero + one LM iteration
    ctx.close()
    ctx = make()
    t0 = time.perf_counter()
    s = ctx.solve()
    wall = time.perf_counter() - t0
    ctx.close()
    # one more solve with HIP events around the phases of every iteration (kept out of the timed solve above)
    phases = None
    try:
        os.environ["MVGX_BA_PHASE_TIMING"] = "1"
        ctx = make()
        sp = ctx.solve()
        ctx.close()
        it = max(sp.num_iterations, 1)
        phases = {"jacobian_ms": sp.jacobian_ms / (sp.num_successful_steps + 1), "schur_ms": sp.schur_ms / it, "solve_ms": sp.solve_ms / it,
                  "backsub_ms": sp.backsub_ms / it, "cost_ms": sp.cost_ms / it,
                  "note": "device time per LM iteration (jacobian: per Jacobian evaluation); the rest of an iteration is host round trips"}
    finally:
        os.environ.pop("MVGX_BA_PHASE_TIMING", None)
    scene = full
    n_cols = 6 * scene["n_poses"] + 8 * scene["n_intrinsics"]
    bytes_it = algorithmic_bytes_per_iteration(scene["n_obs"], scene["n_points"], scene["n_poses"], n_cols)
    rec = {
        "config": f"{cfg['n_cams']} cams ({'pinhole' if cfg['model'] == 1 else 'pinhole+K3'}, {cfg['n_intr_groups']} shared "
                  f"intrinsic group(s)), {cfg['n_points']} points, {full['n_obs']} observations, ADJUST_ALL, Huber(16); "
                  f"points sharded over {world} GPU(s)" + (", RCCL all-reduce of the reduced camera system" if world > 1 else ""),
        "lm_iteration_ms": s.iter_ms_mean,
        "first_call_ms_iteration_zero_plus_one_iteration": s1.total_ms,
        "iterations": s.num_iterations, "successful_steps": s.num_successful_steps, "termination": s.termination,
        "solve_ms": s.total_ms, "solve_wall_ms": wall * 1e3, "create_s_host_structure_plus_upload": create_s,
        "phases": phases,
        "reduced_solve": (None if not phases or not phases["solve_ms"] else
                          {"n": n_cols, "flop": n_cols ** 3 / 3.0, "ms": phases["solve_ms"],
                           "achieved_tflops": n_cols ** 3 / 3.0 / (phases["solve_ms"] * 1e-3) / 1e12, "peak_tflops_fp64_mfma": 78.6}),
        "initial_rmse": s.initial_rmse, "final_rmse": s.final_rmse, "final_cost": s.final_cost,
        "roofline": {"bound": "hbm", "achieved": bytes_it / (s.iter_ms_mean * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": bytes_it / (s.iter_ms_mean * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_iteration": bytes_it},
    }
    if cpu and world == 1:
        try:
            from tests import _oracle
            if _oracle.have_ref_ba():
                t0 = time.perf_counter()
                report, (rc, st, *_) = _capture_stderr(lambda: _oracle.ref_ba_adjust(scene, max_iterations=1, print_summary=1))
                t1 = time.perf_counter() - t0
                c = {"kind": "reference", "cores": os.cpu_count(), "one_iteration_adjust_s": st[2], "wall_s": t1,
                     "ceres_full_report": _parse_report(report),
                     "sample": "Bundle_Adjustment_Ceres::Adjust (vendored Ceres 1.13, SPARSE_SCHUR/EIGEN_SPARSE, OpenMP), "
                               "same scene, max_num_iterations=1 (problem build + preprocessing + iteration 0 + 1 LM iteration)"}
                if st[2] * 12 < cpu_budget_s:   # full solve only when it fits the budget
                    rc, st2, *_ = _oracle.ref_ba_adjust(scene)
                    c["full_solve_s"] = st2[2]; c["final_rmse"] = st2[1]
                    c["rmse_diff_vs_gpu"] = abs(st2[1] - s.final_rmse)
                rec["cpu_baseline"] = c
        except Exception as e:  # side figure only
            rec["cpu_baseline"] = {"kind": "reference", "error": repr(e)}
    return rec


if __name__ == "__main__":
    import json
    import sys
    name = sys.argv[1] if len(sys.argv) > 1 else None
    print(json.dumps(ba_bench_record(0, 1, cpu=(name is None and "--no-cpu" not in sys.argv), name=name)))
