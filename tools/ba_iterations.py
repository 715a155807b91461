"""N LM iterations on a bench scene (c3 | c5) - the command profiled by the rocprofv3 passes of the BA solver.
Usage: ba_iterations.py [c3|c5] [iterations] [--warm]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_ba
from openmvg_amd import ba, synth
name = sys.argv[1] if len(sys.argv) > 1 else "c5"
its = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sc = synth.ba_scene(**bench_ba.ba_config(1, None if name == "c3" else "c5"))
if "--warm" in sys.argv:   # a throw-away solve first: kernels loaded, caches filled (what bench_ba.py times)
    c = ba.BaContext(sc); c.solve(ba.default_options(max_num_iterations=its)); c.close()
c = ba.BaContext(sc)
s = c.solve(ba.default_options(max_num_iterations=its))
print(name, "iterations", s.num_iterations, "iter_ms", s.iter_ms_mean, "rmse", repr(s.final_rmse),
      *(("solve_ms/it", s.solve_ms / max(s.num_iterations, 1), "schur_ms/it", s.schur_ms / max(s.num_iterations, 1), "backsub_ms/it", s.backsub_ms / max(s.num_iterations, 1),
         "jacobian_ms/eval", s.jacobian_ms / (s.num_successful_steps + 1)) if os.environ.get("MVGX_BA_PHASE_TIMING") else ()))
c.close()
