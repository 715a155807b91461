# rocprofv3 --kernel-trace --stats of one pass of every geometric-filter model (tools/geofilter_run.py 20000 250 <m>): the estimation
# kernel's duration per model -> gpurun_out/$CALL/geofilter_kernel_stats.txt.   gpurun -- 'CALL=r4_64 bash tools/geofilter_models_prof.sh'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${CALL:-r4_xx}; mkdir -p $O; rm -f $O/geofilter_kernel_stats.txt; cd /tmp; export TMPDIR=/tmp
for m in f h e a u; do
  (timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -o g -- python $R/tools/geofilter_run.py 20000 250 $m > $O/run_$m.log 2>&1)
  f=$(find $O/prof_$m -name "*kernel_stats.csv" | head -1)
  echo "## model $m: $(grep '^model' $O/run_$m.log)" >> $O/geofilter_kernel_stats.txt
  python3 - "$f" >> $O/geofilter_kernel_stats.txt <<'PY'
import csv, re, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:3]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    print(f"  {name}: calls {r['Calls']}, average {float(r['AverageNs']) / 1e6:.3f} ms, {float(r['Percentage']):.2f} % of the device time")
PY
  rm -rf $O/prof_$m
done
cat $O/geofilter_kernel_stats.txt
