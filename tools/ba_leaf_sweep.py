"""Sweep of the nested-dissection leaf width (MVGX_BA_ND_LEAF_COLS, default 192) on the bench scenes: iteration time, solve phase and the plan's
shape (levels, factor tiles). Usage: ba_leaf_sweep.py [c3|c5] leaf [leaf ...]   (one process per value: the variable is read at create)"""
import os, subprocess, sys
if len(sys.argv) > 2 and sys.argv[1] != "--one":
    for leaf in sys.argv[2:]:
        env = dict(os.environ, MVGX_BA_ND_LEAF_COLS=leaf, MVGX_BA_PHASE_TIMING="1")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one", sys.argv[1], leaf], env=env)
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_ba
from openmvg_amd import ba, synth
name, leaf = sys.argv[2], sys.argv[3]
sc = synth.ba_scene(**bench_ba.ba_config(1, None if name == "c3" else "c5"))
its = 6
c = ba.BaContext(sc); c.solve(ba.default_options(max_num_iterations=its)); c.close()
best = None
for _ in range(3):
    c = ba.BaContext(sc)
    s = c.solve(ba.default_options(max_num_iterations=its))
    info = c.solver_info()
    row = (s.iter_ms_mean, s.solve_ms / max(s.num_iterations, 1), s.schur_ms / max(s.num_iterations, 1), s.num_iterations, s.final_rmse)
    best = row if best is None or row[0] < best[0] else best
    c.close()
print(f"{name} leaf {leaf}: iter_ms {best[0]:.4f} solve_ms/it {best[1]:.4f} schur_ms/it {best[2]:.4f} iterations {best[3]} rmse {best[4]:.9f} | sparse {info.sparse} parts {info.n_parts} "
      f"levels {info.n_levels} factor_tiles {info.n_factor_tiles} dense_tiles {info.n_dense_tiles} padded {info.n_padded}", flush=True)
