#!/bin/bash
# kernel timeline of one LM iteration (c5, c3), no counters
mkdir -p gpurun_out/${CALL_DIR:-r3_50}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/${CALL_DIR:-r3_50}
R=$GRAFT_REPO_ROOT
for s in c5 c3; do
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$s -o ba -- python $R/tools/ba_iterations.py $s 4 --warm > $O/prof_$s.log 2>&1)
  T=$(find $O/prof_$s -name "*kernel_trace.csv" | head -1); python tools/ba_timeline.py $T > $O/ba_${s}_iteration_timeline.txt 2>&1
  rm -rf $O/prof_$s
  grep -v "sp_\(factor\|gemm\|backsolve\)" $O/ba_${s}_iteration_timeline.txt
done
