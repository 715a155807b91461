#!/usr/bin/env python
"""SQ counters of the geometric-filter kernel (two rocprofv3 passes per model, summaries of tools/pmc_kernels.py under <dir>) ->
one record per model: waves per SIMD, the split of the wave cycles (busy issuing / issue stalls / parked), VALU utilisation and the
instruction mix per a-contrario iteration.  geofilter_pmc_summary.py <dir> > (prints, and writes <dir>/geofilter_pmc_summary.json)"""
import json, os, re, sys
d = sys.argv[1]
out = {}
for m in "fhe":
    try:
        a = json.load(open(os.path.join(d, f"geofilter_{m}_pmc_a.json"))); b = json.load(open(os.path.join(d, f"geofilter_{m}_pmc_b.json")))
    except OSError as e:
        print(m, "missing", e); continue
    ka = [k for k in a["per_kernel"] if k.startswith("geofilter_f_acransac_kernel")][0]
    A, B = a["per_kernel"][ka], b["per_kernel"][ka]
    log = open(os.path.join(d, f"pmc_geo_a_{m}.log")).read()
    mm = re.search(r"kernel_ms ([0-9.]+) ok (\d+) iterations (\d+) models (\d+) wave_clocks (\d+)", log)
    its = int(mm.group(3)) if mm else None
    cyc = A["GRBM_GUI_ACTIVE"] / 8.0   # kernel cycles per XCD (all kernels of the run: the filter kernel is > 95 % of them)
    wave_cycles = A["SQ_WAVE_CYCLES"] * 4.0   # quad-cycles -> cycles (MI355X_MICROARCH.md)
    rec = {"kernel": ka, "command_a": a.get("command"), "command_b": b.get("command"), "dispatches": A["dispatches"],
           "iterations": its, "models": int(mm.group(4)) if mm else None, "kernel_ms_under_pmc": float(mm.group(1)) if mm else None,
           "waves_per_simd_mean": A["SQ_WAVE_CYCLES"] / max(A["SQ_BUSY_CYCLES"], 1.0) / 4.0 if "SQ_BUSY_CYCLES" in A else None,
           "wave_cycles_split": {"issue_stall_WAIT_INST_ANY": A["SQ_WAIT_INST_ANY"] / A["SQ_WAVE_CYCLES"], "parked_WAIT_ANY": A["SQ_WAIT_ANY"] / A["SQ_WAVE_CYCLES"]},
           "valu_active_share_of_wave_cycles": B["SQ_ACTIVE_INST_VALU"] / max(A["SQ_WAVE_CYCLES"], 1.0),
           "per_iteration": ({"wave_cycles": wave_cycles / its, "valu": B["SQ_INSTS_VALU"] / its, "salu": B["SQ_INSTS_SALU"] / its, "lds": B["SQ_INSTS_LDS"] / its,
                              "vmem_rd": B["SQ_INSTS_VMEM_RD"] / its} if its else None),
           "raw": {"a": A, "b": B}}
    out[m] = rec
    print(m, json.dumps({k: v for k, v in rec.items() if k not in ("raw", "command_a", "command_b")}))
json.dump(out, open(os.path.join(d, "geofilter_pmc_summary.json"), "w"), indent=1)
