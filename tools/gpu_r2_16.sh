#!/bin/bash
# round 2, call 16: where the time of ba_schur_group_kernel goes (parts switched off, results invalid)
mkdir -p gpurun_out/r2_16
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2_16
for dbg in 0 1 2 4 8 15; do
  MVGX_BA_GROUP_DEBUG=$dbg timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$dbg -o c5 -- python $GRAFT_REPO_ROOT/tools/ba_one_iteration.py > $O/run$dbg.log 2>&1
  python - <<PY
import csv
rows=list(csv.DictReader(open("$O/p$dbg/c5_kernel_stats.csv")))
for r in rows:
    if 'ba_schur_group' in r['Name']: print("dbg $dbg", r['Calls'], '%.1f us' % (float(r['AverageNs'])/1e3))
PY
done
