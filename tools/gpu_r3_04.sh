#!/bin/bash
# round 3, call 4: SQ counters of the BA kernels (instruction mix, wait split, LDS conflicts) on C5
mkdir -p gpurun_out/r3_04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r3_04
R=$GRAFT_REPO_ROOT
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p1 -o m -- python $R/tools/ba_iterations.py c5 3 > $O/p1.log 2>&1)
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY --output-format csv -d $O/p2 -o m -- python $R/tools/ba_iterations.py c5 3 > $O/p2.log 2>&1)
python tools/pmc_kernels.py $O/p1 --window ba_cam_gram_kernel > $O/sq1.json 2> $O/e1
python tools/pmc_kernels.py $O/p2 --window ba_cam_gram_kernel > $O/sq2.json 2> $O/e2
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3_04/"
a=json.load(open(O+"sq1.json"))["per_kernel"]; b=json.load(open(O+"sq2.json"))["per_kernel"]
for k in a:
    if "group" in k or "gram" in k or "linearize" in k:
        x=a[k]; y=b.get(k,{})
        w=max(x["SQ_WAVES"],1)
        print(k[:40], "waves",int(w), "per wave: valu %.0f mfma %.0f lds %.0f salu %.0f vmem %.0f | wave_cycles %.0f (x4 clk)"%(x["SQ_INSTS_VALU"]/w,x["SQ_INSTS_MFMA"]/w,x["SQ_INSTS_LDS"]/w,x["SQ_INSTS_SALU"]/w,x["SQ_INSTS_VMEM"]/w,x["SQ_WAVE_CYCLES"]/w))
        if y:
            wc=a[k]["SQ_WAVE_CYCLES"]
            print("    active_valu %.3f active_lds %.3f wait_any %.3f wait_inst_any %.3f wait_inst_lds %.3f active_any %.3f | bank_conflict/idx_active %.3f"%(
              y["SQ_ACTIVE_INST_VALU"]/wc,y["SQ_ACTIVE_INST_LDS"]/wc,y["SQ_WAIT_ANY"]/wc,y["SQ_WAIT_INST_ANY"]/wc,y["SQ_WAIT_INST_LDS"]/wc,y["SQ_ACTIVE_INST_ANY"]/wc,y["SQ_LDS_BANK_CONFLICT"]/max(y["SQ_LDS_IDX_ACTIVE"],1)))
PY
rm -rf $O/p1 $O/p2
