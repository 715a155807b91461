#!/bin/bash
# round 2, call 4: group kernel with batched staging; adapter delivery diagnosis; streaming RSS at 3000 images; Adjust() latency by size
mkdir -p gpurun_out/r2_04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_04
timeout 600 python -m pytest tests/test_ba_gpu.py -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 300 python bench_ba.py c3x --no-cpu > $O/ba_c3.json 2> $O/ba_c3.err
timeout 300 python bench_ba.py c5 --no-cpu > $O/ba_c5.json 2> $O/ba_c5.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2_04/ba_c*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, r["lm_iteration_ms"], r["iterations"], r["final_rmse"], r["create_s_host_structure_plus_upload"], r["phases"])
    except Exception as e: print(f, "ERR", e)
PY
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c5 -o c5 -- python $GRAFT_REPO_ROOT/bench_ba.py c5 --no-cpu > $GRAFT_REPO_ROOT/$O/prof_c5.log 2>&1)
head -12 $O/prof_c5/c5_kernel_stats.csv | cut -c1-150
for skip in 1 2 0; do
  echo "== adapter, MVGX_ADAPTER_DEBUG_SKIP=$skip"
  MVGX_ADAPTER_DEBUG_SKIP=$skip MVGX_ADAPTER_TIMING=1 timeout 600 python tools/time_adapter_match.py > $O/adapter_match_skip$skip.log 2>&1; grep "^replacement" $O/adapter_match_skip$skip.log
done
timeout 900 python tools/stream_rss.py 3000 > $O/stream_rss_3000.json 2> $O/stream_rss_3000.err; cat $O/stream_rss_3000.json
timeout 900 python tools/time_adapter_ba_sizes.py > $O/adjust_sizes.jsonl 2> $O/adjust_sizes.err; cat $O/adjust_sizes.jsonl
timeout 600 python tools/ba_dense_visibility.py > $O/ba_dense_visibility.json 2> $O/ba_dense_visibility.err; cat $O/ba_dense_visibility.json
