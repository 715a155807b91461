"""Phase times of mvgx_ba_create (MVGX_BA_CREATE_TIMING=1) on the C3 and C5 scenes + end-to-end Adjust() wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MVGX_BA_CREATE_TIMING"] = "1"
import bench_ba
from openmvg_amd import ba, synth
for name in (None, "c5"):
    sc = synth.ba_scene(**bench_ba.ba_config(1, name))
    for rep in range(2):
        print(f"--- {name or 'c3'} rep {rep}", file=sys.stderr, flush=True)
        t = time.perf_counter(); ctx = ba.BaContext(sc, device=0); t1 = time.perf_counter()
        s = ctx.solve(); t2 = time.perf_counter(); ctx.read_params(); ctx.close(); t3 = time.perf_counter()
        print(f"{name or 'c3'}: create {1e3*(t1-t):.1f} ms  solve {1e3*(t2-t1):.1f} ms ({s.num_iterations} it)  read+close {1e3*(t3-t2):.1f} ms", file=sys.stderr, flush=True)
