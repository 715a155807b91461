#!/bin/bash
# launch order of the supergroups: largest first (default) against the order of construction (MVGX_BA_SG_INDEX_ORDER=1), one box
O=$GRAFT_REPO_ROOT/gpurun_out/${CALL_DIR:-r3_85}; mkdir -p $O
for rep in 1 2 3; do
  for s in c3 c5; do
    echo "largest-first $(python tools/ba_iterations.py $s 8 --warm 2>&1 | tail -1)" | tee -a $O/sg_order_ab.txt
    echo "index-order   $(MVGX_BA_SG_INDEX_ORDER=1 python tools/ba_iterations.py $s 8 --warm 2>&1 | tail -1)" | tee -a $O/sg_order_ab.txt
  done
done
timeout 900 python -m pytest tests/test_ba_gpu.py -x -q 2>&1 | tail -2 | tee -a $O/sg_order_ab.txt
