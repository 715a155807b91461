#!/bin/bash
# round 2, call 3: point-group Schur products on the f64 matrix cores; the multi-device tests with the rebuilt adapter
mkdir -p gpurun_out/r2_03
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_03
timeout 1200 python -m pytest tests/test_ba_gpu.py tests/test_adapter_gpu.py tests/test_ba_multirank_gpu.py -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.log
for g in 1 0; do
  MVGX_BA_GROUPS=$g timeout 300 python bench_ba.py c3x --no-cpu > $O/ba_c3_groups$g.json 2> $O/ba_c3_groups$g.err
  MVGX_BA_GROUPS=$g timeout 300 python bench_ba.py c5 --no-cpu > $O/ba_c5_groups$g.json 2> $O/ba_c5_groups$g.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2_03/ba_c*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, r["lm_iteration_ms"], r["iterations"], r["final_rmse"], r["create_s_host_structure_plus_upload"], r["phases"])
    except Exception as e: print(f, "ERR", e)
PY
MVGX_ADAPTER_TIMING=1 timeout 600 python tools/time_adapter_match.py > $O/adapter_match.log 2>&1; grep -v "^\[mvgx" $O/adapter_match.log | tail -5
MVGX_DEVICES=0,0 timeout 600 python tools/time_adapter_match.py > $O/adapter_match_2ctx.log 2>&1; grep -v "^\[mvgx" $O/adapter_match_2ctx.log | tail -5
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c5 -o c5 -- python $GRAFT_REPO_ROOT/bench_ba.py c5 --no-cpu > $GRAFT_REPO_ROOT/$O/prof_c5.log 2>&1
cd $GRAFT_REPO_ROOT; ls $O/prof_c5 | head
