#!/bin/bash
# round 3, call 5: C3 / C5 kernel timelines + full BA GPU tests after the point-group rework
mkdir -p gpurun_out/r3_05
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r3_05
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_ba_multirank_gpu.py tests/test_adapter_gpu.py -m gpu -q -x > $O/pytest_ba.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_ba.log
for S in c3 c5; do
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$S -o ba -- python $R/tools/ba_iterations.py $S 4 > $O/prof_$S.log 2>&1)
T=$(find $O/prof_$S -name "*kernel_trace.csv" | head -1); python tools/ba_timeline.py $T 1 > $O/ba_${S}_iteration_timeline.txt 2>&1
find $O/prof_$S -name "*kernel_stats.csv" -exec cp {} $O/ba_${S}_kernel_stats.csv \;
rm -rf $O/prof_$S
done
cat $O/ba_c3_iteration_timeline.txt
