"""Times the phases of a small matching call (context create / set_regions / run / destroy) — looks for per-call overheads."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmvg_amd import matching, synth

descs = synth.image_descriptors(6, n_desc=300, seed=5)
pairs = matching.exhaustive_pairs_array(6)
for ov in (1, 0, 1):
    for bp in (0, 2):
        t = [time.perf_counter()]
        ctx = matching.MatchContext(0); t.append(time.perf_counter())
        ctx.set_option("overlap", ov)
        if bp: ctx.set_option("batch_pairs", bp)
        ctx.set_regions(descs); t.append(time.perf_counter())
        ctx.run(pairs, np.float32(0.64)); t.append(time.perf_counter())
        ctx.run(pairs, np.float32(0.64)); t.append(time.perf_counter())
        ctx.close(); t.append(time.perf_counter())
        print(f"overlap={ov} batch={bp}: create {t[1]-t[0]:.4f} set_regions {t[2]-t[1]:.4f} run1 {t[3]-t[2]:.4f} run2 {t[4]-t[3]:.4f} close {t[5]-t[4]:.4f}", flush=True)
