#!/bin/bash
# round 2, call 17: group-kernel output through LDS
mkdir -p gpurun_out/r2_17
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_17
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_ba_multirank_gpu.py tests/test_adapter_gpu.py -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 300 python bench_ba.py c3x --no-cpu > $O/ba_c3.json 2> $O/ba_c3.err
timeout 300 python bench_ba.py c5 --no-cpu > $O/ba_c5.json 2> $O/ba_c5.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2_17/ba_c*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "%.3f" % r["lm_iteration_ms"], r["iterations"], "%.9f" % r["final_rmse"], r["phases"])
    except Exception as e: print(f, "ERR", e)
PY
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c5 -o c5 -- python $GRAFT_REPO_ROOT/bench_ba.py c5 --no-cpu > $GRAFT_REPO_ROOT/$O/prof_c5.log 2>&1)
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r2_17/prof_c5/c5_kernel_stats.csv")))
for r in rows[:12]:
    print(r['Name'][:64].ljust(64), r['Calls'].rjust(5), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(9), r['Percentage'])
PY
