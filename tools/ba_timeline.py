#!/usr/bin/env python
"""Timeline of ONE LM iteration from a rocprofv3 --kernel-trace CSV (<prefix>_kernel_trace.csv): every launch between two
consecutive Jacobian evaluations (ba_cam_gram_kernel), with its start offset, duration and the idle gap before it.
Usage: ba_timeline.py <kernel_trace.csv> [which-iteration (default: the last complete one) | head | tail]
head: everything up to the second Jacobian evaluation (iteration zero and the first iteration); tail: from the last one on."""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
marks = [i for i, k in enumerate(ks) if "ba_cam_gram_kernel" in k[2] or "ba_linearize_kernel<true>" in k[2] and False]
if len(marks) < 2:
    sys.exit("fewer than two Jacobian evaluations in the trace")
arg = sys.argv[2] if len(sys.argv) > 2 else str(len(marks) - 2)
if arg == "head":
    which, a, b = "head", 0, marks[1]
elif arg == "tail":
    which, a, b = "tail", marks[-1], len(ks) - 1
else:
    which = int(arg)
    a, b = marks[which], marks[which + 1]
t0 = ks[a][0]
prev_end = t0
busy = 0
gaps = []
print(f"iteration {which}: {b - a} launches, {(ks[b][0] - t0) / 1e3:.1f} us from Jacobian evaluation to Jacobian evaluation")
for s, e, name in ks[a:b]:
    short = name.replace("(anonymous namespace)::", "").split("(")[0][:48]
    gap = (s - prev_end) / 1e3
    print(f"  +{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  gap {gap:7.1f}  {short}")
    busy += e - s
    if gap > 8.0: gaps.append((gap, short))
    prev_end = max(prev_end, e)
tail = (ks[b][0] - prev_end) / 1e3
print(f"  gap before the next Jacobian evaluation {tail:.1f} us")
print(f"busy {busy / 1e3:.1f} us, idle {((ks[b][0] - t0) - busy) / 1e3:.1f} us; gaps > 8 us: " + ", ".join(f"{g:.0f} us before {n}" for g, n in gaps) + (f", {tail:.0f} us at the end" if tail > 8 else ""))
