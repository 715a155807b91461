#!/bin/bash
# same-box A/B of the BA iteration: the tree's library against tools/_build/libmvgx_prev.so (built from HEAD), alternating
O=gpurun_out/${CALL_DIR:-r3_60}; mkdir -p $O
for rep in 1 2 3; do
  for s in c3 c5; do
    echo "new  $(python tools/ba_iterations.py $s 8 --warm 2>&1 | tail -1)" | tee -a $O/ab.txt
    echo "prev $(MVGX_LIB_PATH=$PWD/tools/_build/libmvgx_prev.so python tools/ba_iterations.py $s 8 --warm 2>&1 | tail -1)" | tee -a $O/ab.txt
  done
done
timeout 900 python -m pytest tests/test_ba_gpu.py -x -q 2>&1 | tail -3 | tee -a $O/ab.txt
