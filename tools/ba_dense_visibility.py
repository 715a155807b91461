"""Dense-visibility stress scene for the reduced-system solver choice ("a dense-visibility stress scene showing
no regression"): 200 cameras, every point seen by 100 consecutive cameras -> the reduced camera system is (almost) full. The
automatic choice must be the dense blocked Cholesky and must not be slower than it was; the block-sparse solver forced onto the
same scene is reported beside it. Prints one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmvg_amd import ba, synth

sc = synth.ba_scene(n_cams=200, n_points=4000, track_len=100, model=1, n_intr_groups=1, seed=0xDE75E)
out = {"scene": "200 cams, 4000 points, 100 consecutive cameras per point, %d observations" % sc["n_obs"]}
for mode in ("auto", "dense", "sparse"):
    if mode == "auto":
        os.environ.pop("MVGX_BA_SOLVER", None)
    else:
        os.environ["MVGX_BA_SOLVER"] = mode
    os.environ["MVGX_BA_PHASE_TIMING"] = "1"
    try:
        c = ba.BaContext(sc)
        s = c.solve()
        info = c.solver_info()
        c.close()
        it = max(s.num_iterations, 1)
        out[mode] = {"solver": "sparse" if info.sparse else "dense", "lm_iteration_ms": s.iter_ms_mean, "iterations": s.num_iterations,
                     "final_rmse": s.final_rmse, "reduced_solve_ms": s.solve_ms / it, "levels": info.n_levels,
                     "factor_tiles": info.n_factor_tiles, "dense_tiles": info.n_dense_tiles, "flop": info.flops,
                     "reduced_solve_tflops": info.flops / (s.solve_ms / it * 1e-3) / 1e12 if s.solve_ms else None}
    except Exception as e:
        out[mode] = {"error": repr(e)}
print(json.dumps(out))
