#!/bin/bash
# round 3, call 3: phase stamps of the point-group kernel (MVGX_BA_GROUP_DEBUG), C3 / C5 timings after the bank-conflict fix
mkdir -p gpurun_out/r3_03
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r3_03
R=$GRAFT_REPO_ROOT
MVGX_BA_GROUP_DEBUG=1 timeout 300 python tools/ba_iterations.py c5 4 > $O/stamps_c5.log 2>&1; tail -3 $O/stamps_c5.log
MVGX_BA_PHASE_TIMING=1 timeout 300 python tools/ba_iterations.py c5 6 > $O/c5.log 2>&1; tail -1 $O/c5.log
MVGX_BA_PHASE_TIMING=1 timeout 300 python tools/ba_iterations.py c3 6 > $O/c3.log 2>&1; tail -1 $O/c3.log
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ba -- python $R/tools/ba_iterations.py c5 4 > $O/prof.log 2>&1)
T=$(find $O/prof -name "*kernel_trace.csv" | head -1); python tools/ba_timeline.py $T 1 > $O/ba_c5_iteration_timeline.txt 2>&1; grep -v "sp_\|rocclr" $O/ba_c5_iteration_timeline.txt | head -40
rm -rf $O/prof
