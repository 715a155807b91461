#!/usr/bin/env python
"""Per-kernel sums of a rocprofv3 --pmc pass (csv: *_counter_collection.csv, optionally *_kernel_trace.csv beside it).

  pmc_kernels.py <dir> [--window '<kernel substring>' [k]] [--note '<command line>']

Without --window: {kernel: {dispatches, <counter>: sum, ...}} over the whole run. With --window: only the dispatches between the
k-th and the (k+1)-th dispatch of the named kernel (default: the last complete window) - for the BA solver the window kernel is
ba_point_group_kernel / ba_linearize_kernel<true>, so a window is exactly one LM iteration.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE shows HALF the bytes
of wide (16 B / lane) coalesced reads, other widths and WRITE_SIZE are uncalibrated - the summary therefore carries the raw bytes and
the doubled reads as an upper bound."""
import csv, glob, json, os, sys
from collections import defaultdict, OrderedDict


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()


def main():
    d = sys.argv[1]
    window = None; which = None; note = None
    a = sys.argv[2:]
    while a:
        if a[0] == "--window":
            window = a[1]; a = a[2:]
            if a and a[0].lstrip("-").isdigit():
                which = int(a[0]); a = a[1:]
        elif a[0] == "--note":
            note = a[1]; a = a[2:]
        else:
            sys.exit("unknown argument " + a[0])
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit("no counter_collection.csv under " + d)
    disp = {}   # dispatch id -> (kernel, {counter: value})
    for f in files:
        for r in csv.DictReader(open(f)):
            k = int(r["Dispatch_Id"])
            e = disp.setdefault(k, (short(r["Kernel_Name"]), defaultdict(float)))
            e[1][r["Counter_Name"]] += float(r["Counter_Value"])
    ids = sorted(disp)
    lo, hi = ids[0], ids[-1] + 1
    out = OrderedDict()
    if note:
        out["command"] = note
    if window:
        marks = [k for k in ids if window in disp[k][0]]
        if len(marks) < 2:
            sys.exit(f"fewer than two dispatches of '{window}'")
        w = which if which is not None else len(marks) - 2
        lo, hi = marks[w], marks[w + 1]
        out["window"] = {"kernel": window, "index": w, "of": len(marks) - 1, "dispatches": sum(1 for k in ids if lo <= k < hi)}
    per = OrderedDict()
    for k in ids:
        if not (lo <= k < hi):
            continue
        name, c = disp[k]
        e = per.setdefault(name, defaultdict(float))
        e["dispatches"] += 1
        for cn, v in c.items():
            e[cn] += v
    tot = defaultdict(float)
    for name, e in per.items():
        for cn, v in e.items():
            tot[cn] += v
    out["per_kernel"] = {k: dict(v) for k, v in per.items()}
    out["total"] = dict(tot)
    if "FETCH_SIZE" in tot:
        out["hbm_read_bytes_raw"] = tot["FETCH_SIZE"] * 1024.0
        out["hbm_read_bytes_x2_upper_bound"] = tot["FETCH_SIZE"] * 2048.0
    if "WRITE_SIZE" in tot:
        out["hbm_write_bytes_raw"] = tot["WRITE_SIZE"] * 1024.0
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
