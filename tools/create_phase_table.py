"""Phase times of mvgx_ba_create over several warm runs (MVGX_BA_CREATE_TIMING=1), one column per run.
Usage: create_phase_table.py [c3|c5] [runs]"""
import os, re, subprocess, sys
name = sys.argv[1] if len(sys.argv) > 1 else "c5"
runs_n = sys.argv[2] if len(sys.argv) > 2 else "6"
here = os.path.dirname(os.path.abspath(__file__))
out = subprocess.run([sys.executable, os.path.join(here, "time_ba_create.py"), name, runs_n], capture_output=True, text=True).stderr
runs = []; cur = {}
for l in out.splitlines():
    m = re.match(r"\[mvgx_ba_create\] (.*?)\s+([\d.]+) ms", l)
    if m:
        cur[m.group(1).strip()] = float(m.group(2))
    elif "create total" in l:
        cur["python wall"] = float(l.split()[-2]); runs.append(cur); cur = {}
for k in runs[-1]:
    print("%-40s" % k, " ".join("%7.2f" % r.get(k, 0) for r in runs[1:]))
