#!/bin/bash
# wave-level counters of one C5 / C3 LM iteration on the final kernels (two passes: no sys / hip trace domains beside --pmc)
O=$GRAFT_REPO_ROOT/gpurun_out/${CALL_DIR:-r3_82}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
for s in c5 c3; do
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/pmc_a_$s -o m -- python $R/tools/ba_iterations.py $s 3 --warm > $O/pmc_a_$s.log 2>&1)
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/pmc_b_$s -o m -- python $R/tools/ba_iterations.py $s 3 --warm > $O/pmc_b_$s.log 2>&1)
  python tools/pmc_kernels.py $O/pmc_a_$s --window 'ba_cam_gram_kernel' --note "rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -- python tools/ba_iterations.py $s 3 --warm" > $O/ba_${s}_pmc_waves_a.json 2> $O/a_$s.err
  python tools/pmc_kernels.py $O/pmc_b_$s --window 'ba_cam_gram_kernel' --note "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVES -- python tools/ba_iterations.py $s 3 --warm" > $O/ba_${s}_pmc_waves_b.json 2> $O/b_$s.err
  tail -2 $O/pmc_a_$s.log; tail -2 $O/a_$s.err $O/b_$s.err
  rm -rf $O/pmc_a_$s $O/pmc_b_$s
done
python - <<'PY'
import json, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/"+os.environ.get("CALL_DIR","r3_82")+"/"
for s in ("c5","c3"):
    try:
        a=json.load(open(O+f"ba_{s}_pmc_waves_a.json")); b=json.load(open(O+f"ba_{s}_pmc_waves_b.json"))
    except Exception as e:
        print(s, "failed", e); continue
    ka=a.get("per_kernel", a); kb=b.get("per_kernel", b)
    for k in ka:
        if "point_group" in k or "gram_kernel" in k:
            print(s, k[:40], {c: ka[k].get(c) for c in ("SQ_WAVE_CYCLES","SQ_WAIT_INST_ANY","SQ_BUSY_CYCLES")}, {c: kb.get(k,{}).get(c) for c in ("SQ_INSTS_VALU","SQ_ACTIVE_INST_VALU","SQ_INSTS_LDS","SQ_WAVES")})
PY
