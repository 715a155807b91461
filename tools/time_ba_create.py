"""Phase times of mvgx_ba_create (MVGX_BA_CREATE_TIMING=1: host structure build vs allocation vs upload) on a bench scene.
Usage: time_ba_create.py [c3|c5] [repeats]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MVGX_BA_CREATE_TIMING"] = "1"
import bench_ba
from openmvg_amd import ba, synth
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sc = synth.ba_scene(**bench_ba.ba_config(1, None if name == "c3" else "c5"))
for rep in range(reps):
    t = time.perf_counter(); c = ba.BaContext(sc); dt = time.perf_counter() - t; c.close()
    print(f"{name} create total {dt * 1e3:.2f} ms", file=sys.stderr)
