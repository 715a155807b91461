# where a whole call of the geometric filter goes (MVGX_GEO_TIMING=1), two calls each (the second with warm caches); SPECS="pairs n model;..." overrides the bench sizes
IFS=';' read -ra specs <<< "${SPECS:-100000 250 f;20000 250 h;20000 250 e}"
for spec in "${specs[@]}"; do
  set -- $spec
  MVGX_GEO_TIMING=1 python - "$@" 2>&1 <<'PY' | grep "mvgx geofilter"
import sys
src = open("tools/geofilter_run.py").read()
exec(src); exec(src)
PY
done
