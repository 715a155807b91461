#!/usr/bin/env python
"""Summarise a rocprofv3 PMC pass (--kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum, csv output)
into per-kernel counter sums and the HBM traffic of the matching filter kernel per launch and per image pair, converted as
/opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: reads = TCC_EA0_RDREQ x 64 B x 2 (gfx950 tallies the 128-B
requests of 16 B/lane streams at 64 B), writes = TCC_EA0_WRREQ x 64 B.
Usage: pmc_traffic_summary.py <dir with *_counter_collection.csv> <image pairs in the run> <command string> > out.json"""
import csv, glob, json, os, sys
from collections import defaultdict

d, n_pairs, command = sys.argv[1], float(sys.argv[2]), sys.argv[3]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
if not files:
    sys.exit("no counter_collection.csv under " + d)
sums = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        sums[name][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[name].add(r["Dispatch_Id"])
out = {"command": command, "image_pairs": n_pairs,
       "per_kernel_sums": {k: dict(v, _dispatches=len(disp[k])) for k, v in sums.items()}}
fk = [k for k in sums if k.startswith("l2_filter")]   # l2_filter_kernel (32x32x32) or l2_filter16_kernel (16x16x64, round 5)
if fk:
    s = sums[fk[0]]
    rd, wr = s["TCC_EA0_RDREQ_sum"] * 64.0 * 2.0, s["TCC_EA0_WRREQ_sum"] * 64.0
    n = len(disp[fk[0]])
    out["filter_kernel"] = {"launches": n, "hbm_read_bytes_with_gfx950_x2_correction": rd, "hbm_write_bytes": wr,
                            "bytes_per_launch": (rd + wr) / max(n, 1), "bytes_per_image_pair": (rd + wr) / n_pairs,
                            "l2_hit_rate": s["TCC_HIT_sum"] / max(s["TCC_HIT_sum"] + s["TCC_MISS_sum"], 1.0)}
json.dump(out, sys.stdout, indent=1)
print()
