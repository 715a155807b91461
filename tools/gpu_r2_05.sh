#!/bin/bash
# round 2, call 5: adapter with the asynchronous list builder (stream_hold); Adjust() latency by size with the device-memory arena
mkdir -p gpurun_out/r2_05
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_05
timeout 900 python -m pytest tests/test_adapter_gpu.py tests/test_matching_gpu.py tests/test_real_images.py tests/test_ba_gpu.py tests/test_ba_multirank_gpu.py -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.log
for skip in 1 2 0; do
  echo "== adapter, MVGX_ADAPTER_DEBUG_SKIP=$skip"
  MVGX_ADAPTER_DEBUG_SKIP=$skip MVGX_ADAPTER_TIMING=1 timeout 600 python tools/time_adapter_match.py > $O/adapter_match_skip$skip.log 2>&1; grep "^replacement" $O/adapter_match_skip$skip.log
done
echo "== adapter, pinned stream buffers"; MVGX_ADAPTER_PINNED_RESULTS=1 timeout 600 python tools/time_adapter_match.py > $O/adapter_match_pinned.log 2>&1; grep "^replacement" $O/adapter_match_pinned.log
echo "== adapter, two contexts"; MVGX_DEVICES=0,0 timeout 600 python tools/time_adapter_match.py > $O/adapter_match_2ctx.log 2>&1; grep "^replacement" $O/adapter_match_2ctx.log
timeout 1200 python tools/time_adapter_ba_sizes.py > $O/adjust_sizes.jsonl 2> $O/adjust_sizes.err; cat $O/adjust_sizes.jsonl
MVGX_BA_CREATE_TIMING=1 python - > $O/create_timing_small.log 2>&1 <<'PY'
import time
from openmvg_amd import ba, synth
for n_cams, n_pts, tl in ((3, 1500, 3), (50, 20000, 6)):
    sc = synth.ba_scene(n_cams=n_cams, n_points=n_pts, track_len=tl, model=3, n_intr_groups=1, seed=1)
    for rep in range(3):
        t0 = time.perf_counter(); c = ba.BaContext(sc); t1 = time.perf_counter(); s = c.solve(); t2 = time.perf_counter(); c.read_params(); c.close(); t3 = time.perf_counter()
        print(f"### {n_cams} views rep {rep}: create {1e3*(t1-t0):.2f} ms, solve {1e3*(t2-t1):.2f} ms ({s.num_iterations} iterations, device {s.total_ms:.2f} ms), read+close {1e3*(t3-t2):.2f} ms", flush=True)
PY
grep "###" $O/create_timing_small.log
