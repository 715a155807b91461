#!/bin/bash
# round 3, call 9: group size of the fused point-group pass (observations per workgroup): 128 / 192 / 256
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for t in 128 192 256; do
  echo "== $t threads per group"
  MVGX_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/libmvgx_g$t.so python tools/ba_iterations.py c5 6 2>&1 | tail -1
  MVGX_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/libmvgx_g$t.so python tools/ba_iterations.py c3 6 2>&1 | tail -1
  (cd /tmp; MVGX_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/libmvgx_g$t.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$t -o ba -- python $GRAFT_REPO_ROOT/tools/ba_iterations.py c5 4 > /tmp/prof_$t.log 2>&1)
  T=$(find /tmp/prof_$t -name "*kernel_trace.csv" | head -1); python tools/ba_timeline.py $T 1 2>&1 | grep "group_kernel\|cam_gram\|launches"
done
