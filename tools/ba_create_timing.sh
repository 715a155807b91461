mkdir -p gpurun_out/${CALL:-r6_41}
MVGX_BA_CREATE_TIMING=1 python - > gpurun_out/${CALL:-r6_41}/create_timing.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, '.')
import bench_ba
from openmvg_amd import ba, synth
for name in ("c3", "c5"):
    sc = synth.ba_scene(**bench_ba.ba_config(1, None if name == "c3" else "c5"))
    for rep in range(4):
        t = time.perf_counter(); c = ba.BaContext(sc); dt = time.perf_counter() - t
        print(f"== {name} rep {rep} BaContext() {dt*1e3:.2f} ms", file=sys.stderr, flush=True)
        if rep == 0: c.solve(ba.default_options(max_num_iterations=2))
        c.close()
PY
tail -70 gpurun_out/${CALL:-r6_41}/create_timing.txt
