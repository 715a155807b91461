# where a whole call of the geometric filter goes (MVGX_GEO_TIMING=1), bench sizes, two calls each (the second with warm caches)
for spec in "100000 250 f" "20000 250 h" "20000 250 e"; do
  set -- $spec
  MVGX_GEO_TIMING=1 python - "$@" 2>&1 <<'PY' | grep "mvgx geofilter"
import sys
src = open("tools/geofilter_run.py").read()
exec(src); exec(src)
PY
done
