#!/bin/bash
# per-iteration table of the kernel trace (c5, c3): is a slow run a slow box or slow iterations?
O=$GRAFT_REPO_ROOT/gpurun_out/${CALL_DIR:-r3_72}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
for s in c5 c3; do
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$s -o ba -- python $R/tools/ba_iterations.py $s 8 --warm > $O/prof_$s.log 2>&1)
  T=$(find $O/prof_$s -name "*kernel_trace.csv" | head -1); python tools/ba_iteration_table.py $T | tee $O/ba_${s}_iteration_table.txt
  rm -rf $O/prof_$s
done
