# same-box A/B of the geometric-filter kernels: samples ahead (4) against one minimal solve per iteration (1), for the models given
# (default f h e); MVGX_LIB_B = a second library (e.g. another occupancy) run beside the product's
O=gpurun_out/${CALL:-geo_ab}; mkdir -p $O
for rep in 1 2 3; do
  for m in ${MODELS:-f h e}; do
    for a in 4 1; do
      echo -n "$m ahead $a: " >> $O/ab.txt
      MVGX_GEO_AHEAD=$a python tools/geofilter_run.py 20000 250 $m 2>&1 | head -1 >> $O/ab.txt
    done
    if [ -n "$MVGX_LIB_B" ]; then
      echo -n "$m ahead 4 ($MVGX_LIB_B): " >> $O/ab.txt
      MVGX_LIB_PATH=$MVGX_LIB_B MVGX_GEO_AHEAD=4 python tools/geofilter_run.py 20000 250 $m 2>&1 | head -1 >> $O/ab.txt
    fi
  done
done
cat $O/ab.txt
