#!/bin/bash
# round 2, call 21: factor-and-invert kernel with the inverse overlapped on the idle waves + batched pivot-column fetches
mkdir -p gpurun_out/r2_21
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_21
MVGX_BA_FACTOR_DEBUG=1 timeout 300 python tools/ba_one_iteration.py > $O/stamps.log 2>&1; grep "factor kernel" $O/stamps.log
MVGX_BA_FACTOR_DEBUG=1 MVGX_BA_SOLVER=dense timeout 300 python tools/ba_one_iteration.py > $O/stamps_dense.log 2>&1; grep "factor kernel" $O/stamps_dense.log
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_ba_multirank_gpu.py -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 300 python bench_ba.py c3x --no-cpu > $O/ba_c3.json 2> $O/ba_c3.err
timeout 300 python bench_ba.py c5 --no-cpu > $O/ba_c5.json 2> $O/ba_c5.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2_21/ba_c*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "%.3f" % r["lm_iteration_ms"], r["iterations"], "%.9f" % r["final_rmse"], r["phases"])
    except Exception as e: print(f, "ERR", e)
PY
