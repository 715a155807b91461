#!/bin/bash
# round 2, call 6: full GPU suite after the BA changes (model cost from the normal equations, arena, grouped Schur products), bench line
mkdir -p gpurun_out/r2_06
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_06
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu_all.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest_gpu_all.log
timeout 300 python bench_ba.py c3x --no-cpu > $O/ba_c3.json 2> $O/ba_c3.err
timeout 300 python bench_ba.py c5 --no-cpu > $O/ba_c5.json 2> $O/ba_c5.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2_06/ba_c*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, r["lm_iteration_ms"], r["iterations"], r["final_rmse"], r["create_s_host_structure_plus_upload"], r["phases"])
    except Exception as e: print(f, "ERR", e)
PY
timeout 600 python tools/time_adapter_ba_sizes.py --no-ref > $O/adjust_sizes_noref.jsonl 2> $O/adjust_sizes.err; cat $O/adjust_sizes_noref.jsonl
timeout 300 python tools/time_adapter_ba.py > $O/adapter_ba_c3.log 2>&1; tail -3 $O/adapter_ba_c3.log
(time timeout 1200 python bench.py) > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.json; tail -4 $O/bench.err
