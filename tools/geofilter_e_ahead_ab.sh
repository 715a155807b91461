# same-box A/B of the essential filter kernel: samples ahead (4) / one per iteration (1), 2 or 1 workgroups per CU, the round-4 kernel
O=gpurun_out/${CALL:-e_ab}; mkdir -p $O
python -m pytest tests/test_geofilter_e.py -m gpu -x -q > $O/pytest_e.log 2>&1; tail -3 $O/pytest_e.log
for rep in 1 2 3; do
  for v in "new 4" "new 2" "new 1" "wgs1 4" "old 1"; do
    set -- $v
    case $1 in new) L=openmvg_amd/lib/libmvgx_hip.so ;; wgs1) L=tools/_build/libmvgx_e_wgs1.so ;; old) L=tools/_build/libmvgx_e_old.so ;; esac
    echo -n "$1 ahead $2: " >> $O/ab.txt
    MVGX_LIB_PATH=$L MVGX_GEO_E_AHEAD=$2 python tools/geofilter_run.py 20000 250 e 2>&1 | tail -1 >> $O/ab.txt
  done
done
cat $O/ab.txt
