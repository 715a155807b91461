#!/usr/bin/env python
"""One screen of the numbers of a bench.py line (the last JSON line of the file given)."""
import json, sys
r = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
rf = r["roofline"]
print(f"match {r['value']:.4g} {r['unit']}  {r['ms_per_step']:.1f} ms/step  frac {rf['frac']:.3f}  launch {rf['mean_launch_ms']:.2f} ms  parity {r.get('parity', {}).get('identical')}")
for k in ("ba", "ba_c5_single_gpu"):
    b = r.get(k) or {}
    if "lm_iteration_ms" in b:
        print(f"{k}: {b['lm_iteration_ms']:.3f} ms/iteration  phases {b.get('phases')}  rmse diff {b.get('cpu_baseline', {}).get('rmse_diff_vs_reference')}  "
              f"its {b.get('cpu_baseline', {}).get('iterations_gpu_vs_reference')}  traffic x{b.get('roofline', {}).get('traffic_over_algorithmic')}")
    else:
        print(k, b)
for k in ("hamming", "l2_float", "l2_uint8_144"):
    b = r.get(k) or {}
    print(k, f"{b.get('value', 0):.4g}", "frac", b.get("roofline", {}).get("frac"), "parity", b.get("parity"))
for k in ("geometric_filter", "geometric_filter_homography", "geometric_filter_essential"):
    b = r.get(k) or {}
    if "value" in b:
        rf = b.get("roofline", {})
        print(k, f"{b['value']:.4g} pairs/s whole call, kernel {b.get('image_pairs_per_s_kernel_time', 0):.4g}; clocks/iter {rf.get('clocks_per_iteration_and_wave')} "
                 f"alone {rf.get('clocks_per_iteration_one_wave_per_simd')} frac {rf.get('frac')}; parity {b.get('parity')}; cpu {b.get('cpu_baseline', {}).get('value')}")
    elif b:
        print(k, b)
for m, b in (r.get("geometric_filter_other_models") or {}).items():
    if isinstance(b, dict) and "value" in b:
        print(f"geometric_filter -g {m}: {b['value']:.4g} pairs/s whole call, kernel {b.get('image_pairs_per_s_kernel_time', 0):.4g}; parity {b.get('parity')}; "
              f"cpu {b.get('cpu_baseline', {}).get('value')}")
    else:
        print("geometric_filter_other_models", m, b)
