#!/bin/bash
# round 3, call 8: SQ counters of the geometric-filter kernel; create timing after the parallelisation; BA tests
mkdir -p gpurun_out/r3_08
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r3_08
R=$GRAFT_REPO_ROOT
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p1 -o m -- python $R/tools/geofilter_run.py 20000 > $O/p1.log 2>&1)
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH --output-format csv -d $O/p2 -o m -- python $R/tools/geofilter_run.py 20000 > $O/p2.log 2>&1)
python tools/pmc_kernels.py $O/p1 > $O/sq1.json 2> $O/e1; python tools/pmc_kernels.py $O/p2 > $O/sq2.json 2> $O/e2
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3_08/"
a=json.load(open(O+"sq1.json"))["per_kernel"]; b=json.load(open(O+"sq2.json"))["per_kernel"]
for k in a:
    if "geofilter_f" in k:
        x=a[k]; y=b.get(k,{}); w=max(x["SQ_WAVES"],1); wc=x["SQ_WAVE_CYCLES"]
        print(k[:50], "waves",int(w), "per wave: valu %.0f salu %.0f lds %.0f vmem %.0f smem %.0f | wave_cycles(x4 clk) %.0f"%(x["SQ_INSTS_VALU"]/w,x["SQ_INSTS_SALU"]/w,x["SQ_INSTS_LDS"]/w,x["SQ_INSTS_VMEM"]/w,x["SQ_INSTS_SMEM"]/w,wc/w))
        if y: print("   active_valu %.3f active_lds %.3f active_sca %.3f wait_any %.3f wait_inst_any %.3f wait_inst_lds %.3f active_any %.3f branches/wave %.0f"%(y["SQ_ACTIVE_INST_VALU"]/wc,y["SQ_ACTIVE_INST_LDS"]/wc,y["SQ_ACTIVE_INST_SCA"]/wc,y["SQ_WAIT_ANY"]/wc,y["SQ_WAIT_INST_ANY"]/wc,y["SQ_WAIT_INST_LDS"]/wc,y["SQ_ACTIVE_INST_ANY"]/wc,y["SQ_INSTS_BRANCH"]/w))
PY
rm -rf $O/p1 $O/p2
python tools/time_ba_create.py c3 3 2>&1 | tail -11; python tools/time_ba_create.py c5 2 2>&1 | tail -11
timeout 600 python -m pytest tests/test_ba_gpu.py tests/test_adapter_gpu.py -m gpu -q -x 2>&1 | tail -2
