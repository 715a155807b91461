#!/bin/bash
# round 2, call 9: BA after the round-trip removals (kernel stats), cascade layout tests, full GPU suite
mkdir -p gpurun_out/r2_09
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_09
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu_all.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest_gpu_all.log
timeout 300 python bench_ba.py c3x --no-cpu > $O/ba_c3.json 2> $O/ba_c3.err
timeout 300 python bench_ba.py c5 --no-cpu > $O/ba_c5.json 2> $O/ba_c5.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2_09/ba_c*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, r["lm_iteration_ms"], r["iterations"], r["final_rmse"], r["create_s_host_structure_plus_upload"], r["phases"])
    except Exception as e: print(f, "ERR", e)
PY
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c5 -o c5 -- python $GRAFT_REPO_ROOT/bench_ba.py c5 --no-cpu > $GRAFT_REPO_ROOT/$O/prof_c5.log 2>&1)
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench_ba.py c3x --no-cpu > $GRAFT_REPO_ROOT/$O/prof_c3.log 2>&1)
python - <<'PY'
import csv
for tag in ("c5","c3"):
    rows=list(csv.DictReader(open(f"gpurun_out/r2_09/prof_{tag}/{tag}_kernel_stats.csv")))
    print("==", tag)
    for r in rows[:16]:
        print(r['Name'][:64].ljust(64), r['Calls'].rjust(5), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(9), r['Percentage'])
PY
timeout 600 python tools/time_adapter_ba_sizes.py --no-ref > $O/adjust_sizes_noref.jsonl 2> $O/adjust_sizes.err; cat $O/adjust_sizes_noref.jsonl
