#!/bin/bash
# round 2, call 26: state-of-the-tree verification - the full GPU suite, smoke(), the driver's bench command plain and under
# rocprofv3 --kernel-trace --stats, and a PMC pass (HBM traffic) of one pass over the bench workload
mkdir -p gpurun_out/r2_26
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_26
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu_all.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu_all.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
(time timeout 1500 python bench.py) > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r2_26/bench.json").read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['mean_launch_ms'], r['parity']['identical'])
print('ba', r['ba']['lm_iteration_ms'], r['ba']['cpu_baseline'].get('rmse_diff_vs_reference'), 'c5', r['ba_c5_single_gpu']['lm_iteration_ms'], r['ba_c5_single_gpu']['cpu_baseline'].get('rmse_diff_vs_reference'))
print({k: r[k].get('value') for k in ('hamming','l2_float','l2_uint8_144') if k in r})
PY
tail -3 $O/bench.err
(cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-ba --no-hamming > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_under_rocprof.err)
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r2_26/prof_bench/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:8]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(10), r['Percentage'])
PY
CMD="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-ba --no-hamming"
(cd /tmp; timeout 900 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc -o m -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-ba --no-hamming > $GRAFT_REPO_ROOT/$O/pmc.log 2>&1)
python tools/pmc_traffic_summary.py $O/pmc 499500 "rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -- $CMD" > $O/match_traffic_pmc.json 2> $O/pmc_summary.err
python - <<'PY'
import json
try:
    print(json.load(open("gpurun_out/r2_26/match_traffic_pmc.json"))["filter_kernel"])
except Exception as e: print("pmc summary failed", e)
PY
find $O/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
rm -rf $O/pmc $O/prof_bench
