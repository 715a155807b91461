#!/usr/bin/env python
"""MFMA-pipe utilisation and sustained clock of l2_filter_kernel from one rocprofv3 pass
(SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA ... GRBM_GUI_ACTIVE, summary of tools/pmc_kernels.py) + the kernel trace of the same run.
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs
  clock = kernel cycles / kernel time (kernel-trace timestamps); frac of the nominal 5 PFLOP/s = busy x clock / 2.4 GHz
Usage: filter_busy_summary.py <pmc_kernels.json> <kernel_trace.csv>"""
import csv, json, sys
j = json.load(open(sys.argv[1]))
name = [k for k in j["per_kernel"] if k.startswith("l2_filter")][0]   # l2_filter_kernel or l2_filter16_kernel
c = j["per_kernel"][name]
dur_ns = 0; n = 0
for r in csv.DictReader(open(sys.argv[2])):
    if "l2_filter" in r["Kernel_Name"]:
        dur_ns += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); n += 1
cycles = c["GRBM_GUI_ACTIVE"] / 8.0
busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024.0)
clock_ghz = cycles / dur_ns
out = {"kernel": name, "launches": int(c["dispatches"]), "launches_in_trace": n, "mean_launch_ms_under_pmc": dur_ns / max(n, 1) / 1e6,
       "SQ_VALU_MFMA_BUSY_CYCLES": c["SQ_VALU_MFMA_BUSY_CYCLES"], "SQ_INSTS_MFMA": c["SQ_INSTS_MFMA"],
       "busy_cycles_per_mfma": c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(c["SQ_INSTS_MFMA"], 1.0),
       "GRBM_GUI_ACTIVE": c["GRBM_GUI_ACTIVE"], "kernel_cycles_per_xcd": cycles, "mfma_pipe_busy_frac": busy,
       "sustained_clock_ghz": clock_ghz, "nominal_clock_ghz": 2.4, "busy_x_clock_over_nominal": busy * clock_ghz / 2.4,
       "wave_cycles_split": {k: c[k] / c["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") if k in c},
       "valu_per_mfma": c.get("SQ_ACTIVE_INST_VALU", 0.0) / max(c["SQ_INSTS_MFMA"], 1.0),
       "command": j.get("command")}
json.dump(out, sys.stdout, indent=1); print()
