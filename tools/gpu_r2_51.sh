#!/bin/bash
# round 2, call 51: coalesced record reads in the per-point kernels (point norms first): BA tests, iteration times, timeline
mkdir -p gpurun_out/r2_51
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_51
timeout 600 python -m pytest tests/test_ba_gpu.py tests/test_ba_multirank_gpu.py tests/test_adapter_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 300 python bench_ba.py c3x --no-cpu > $O/ba_c3.json 2> $O/ba_c3.err
timeout 300 python bench_ba.py c5 --no-cpu > $O/ba_c5.json 2> $O/ba_c5.err
for c in c3x c5; do
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_$c -o t -- python $GRAFT_REPO_ROOT/bench_ba.py $c --no-cpu > $GRAFT_REPO_ROOT/$O/trace_$c.log 2>&1)
  f=$(find $O/trace_$c -name "*kernel_trace.csv" | head -1)
  python tools/ba_timeline.py $f > $O/timeline_$c.txt 2>&1
  rm -rf $O/trace_$c
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2_51/ba_c*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "%.3f" % r["lm_iteration_ms"], r["iterations"], "%.9f" % r["final_rmse"], r["phases"])
    except Exception as e: print(f, "ERR", e)
PY
grep "point_norms\|point_solve\|slot_z\|backsub_kernel\|obs_z\|model_cost" $O/timeline_c5.txt
