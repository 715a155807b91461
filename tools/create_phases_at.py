"""Phase times (MVGX_BA_CREATE_TIMING=1) of a warm mvgx_ba_create on the scene of tools/adjust_latency_by_size.py with n cameras.
Usage: create_phases_at.py <n_cams> [runs]"""
import os, re, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from openmvg_amd import ba, synth
    n = int(sys.argv[2]); runs = int(sys.argv[3])
    sc = synth.ba_scene(n_cams=n, n_points=500 * n, track_len=min(10, n), model=3, n_intr_groups=min(8, max(1, n // 4)), seed=0xAD705 + n)
    for _ in range(runs):
        c = ba.BaContext(sc); c.close(); print("== run", file=sys.stderr)
    sys.exit(0)
n = sys.argv[1]; runs = sys.argv[2] if len(sys.argv) > 2 else "5"
env = dict(os.environ, MVGX_BA_CREATE_TIMING="1")
out = subprocess.run([sys.executable, __file__, "--child", n, runs], capture_output=True, text=True, env=env).stderr
allr = []; cur = {}
for l in out.splitlines():
    m = re.match(r"\[mvgx_ba_create\] (.*?)\s+([\d.]+) ms", l)
    if m: cur[m.group(1).strip()] = float(m.group(2))
    elif l.startswith("== run"): allr.append(cur); cur = {}
for k in allr[-1]:
    print("%-40s" % k, " ".join("%7.2f" % r.get(k, 0) for r in allr[1:]))
