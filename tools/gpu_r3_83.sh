#!/bin/bash
# supergroup size sweep on one box: MVGX_BA_SG_GROUPS = groups (of up to 256 observations) per supergroup
O=$GRAFT_REPO_ROOT/gpurun_out/${CALL_DIR:-r3_83}; mkdir -p $O
for g in 8 6 12 16 24 8; do
  for s in c3 c5; do
    if [ $g = default ]; then v=$(python tools/ba_iterations.py $s 8 --warm 2>&1 | tail -1); else v=$(MVGX_BA_SG_GROUPS=$g python tools/ba_iterations.py $s 8 --warm 2>&1 | tail -1); fi
    echo "sg_groups=$g $v" | tee -a $O/sg_sweep.txt
  done
done
