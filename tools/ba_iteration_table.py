#!/usr/bin/env python
"""Per LM iteration of a rocprofv3 --kernel-trace CSV: duration from Jacobian evaluation to Jacobian evaluation and the time of
selected kernels inside it (to tell a slow box from a slow iteration). Usage: ba_iteration_table.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
marks = [i for i, k in enumerate(ks) if "ba_cam_gram_kernel" in k[2]]
names = ["sp_factor_kernel", "sp_gemm_kernel<false>", "sp_gemm_kernel<true>", "sp_backsolve", "ba_point_group_kernel<1", "ba_point_group_kernel<2", "ba_cam_gram_kernel"]
print("iteration  total_us  " + "  ".join(n[:22] for n in names))
for w in range(len(marks) - 1):
    a, b = marks[w], marks[w + 1]
    tot = (ks[b][0] - ks[a][0]) / 1e3
    sums = [sum((e - s) for s, e, n in ks[a:b] if nm in n) / 1e3 for nm in names]
    print(f"{w:9d}  {tot:8.1f}  " + "  ".join(f"{v:22.1f}" for v in sums))
