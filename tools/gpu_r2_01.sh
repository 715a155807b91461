#!/bin/bash
# round 2, call 1: GPU tests of the block-sparse reduced solve + config-scale parity tests; BA bench sparse vs dense; full bench line
mkdir -p gpurun_out/r2_01
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_01
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_ba_multirank_gpu.py tests/test_matching_gpu.py -m gpu -x -q > $O/pytest_ba_match.log 2>&1
echo "pytest rc=$?" ; tail -3 $O/pytest_ba_match.log
for mode in sparse dense; do
  MVGX_BA_SOLVER=$mode timeout 300 python bench_ba.py c3x --no-cpu > $O/ba_c3_$mode.json 2> $O/ba_c3_$mode.err
  MVGX_BA_SOLVER=$mode timeout 300 python bench_ba.py c5 --no-cpu > $O/ba_c5_$mode.json 2> $O/ba_c5_$mode.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2_01/ba_c*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, r["lm_iteration_ms"], r["iterations"], r["final_rmse"], r["phases"], (r["reduced_solve"] or {}).get("levels_of_dependent_launches"), (r["reduced_solve"] or {}).get("factor_tiles_64x64"))
    except Exception as e: print(f, "ERR", e)
PY
(time timeout 1200 python bench.py) > $O/bench.json 2> $O/bench.err
tail -c 3000 $O/bench.json; tail -5 $O/bench.err
