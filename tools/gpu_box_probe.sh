#!/bin/bash
# which kind of box is this? BA iteration time, the factor kernel's shader-clock stamps, partition / power settings
python tools/ba_iterations.py c3 8 --warm 2>&1 | tail -1
MVGX_BA_FACTOR_DEBUG=1 python tools/ba_iterations.py c3 3 2>&1 | grep -i "factor kernel" | tail -1
rocm-smi --showcomputepartition --showmemorypartition 2>/dev/null | grep -i "partition" | head -4
rocm-smi --showmaxpower --showperflevel 2>/dev/null | grep -i "power\|level" | head -4
rocminfo 2>/dev/null | grep -i "Max Clock Freq\|Compute Unit" | tail -2
grep -m1 "model name" /proc/cpuinfo
