#!/usr/bin/env python
"""Combines the FETCH_SIZE and WRITE_SIZE window summaries of tools/pmc_kernels.py (one LM iteration each) of the bench scenes into
profiles/round3_ba_iteration_traffic.json, the file bench_ba.py reads for roofline.traffic.
Usage: ba_traffic_from_pmc.py <c3 fetch.json> <c3 write.json> <c5 fetch.json> <c5 write.json> > out.json"""
import json, sys
out = {}
for name, f, w in (("c3", sys.argv[1], sys.argv[2]), ("c5", sys.argv[3], sys.argv[4])):
    jf, jw = json.load(open(f)), json.load(open(w))
    rd, wr = jf["hbm_read_bytes_x2_upper_bound"], jw["hbm_write_bytes_raw"]
    per = {k: {"read_x2": v.get("FETCH_SIZE", 0) * 2048.0, "written": jw["per_kernel"].get(k, {}).get("WRITE_SIZE", 0) * 1024.0, "dispatches": v["dispatches"]}
           for k, v in jf["per_kernel"].items()}
    out[name] = {"hbm_bytes_per_iteration": rd + wr, "read_bytes_x2": rd, "written_bytes": wr, "command": jf.get("command"), "window": jf.get("window"),
                 "per_kernel": dict(sorted(per.items(), key=lambda kv: -(kv[1]["read_x2"] + kv[1]["written"])))}
json.dump(out, sys.stdout, indent=1)
print()
