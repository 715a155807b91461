#!/bin/bash
# round 3, call 11: what a solve spends outside its LM iterations (iteration zero, the final evaluation): head / tail of the c5 and c3 traces
mkdir -p gpurun_out/r3_11
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r3_11
R=$GRAFT_REPO_ROOT
for s in c5 c3; do
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$s -o ba -- python $R/tools/ba_iterations.py $s 6 > $O/prof_$s.log 2>&1)
  T=$(find $O/prof_$s -name "*kernel_trace.csv" | head -1)
  python tools/ba_timeline.py $T head > $O/ba_${s}_head.txt 2>&1
  python tools/ba_timeline.py $T tail > $O/ba_${s}_tail.txt 2>&1
  rm -rf $O/prof_$s
done
cat $O/ba_c5_head.txt | grep -v "sp_\(factor\|gemm\|backsolve\)"; cat $O/ba_c5_tail.txt | grep -v "sp_\(factor\|gemm\|backsolve\)"
