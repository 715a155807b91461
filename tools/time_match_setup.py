#!/usr/bin/env python
"""Where does the time of a one-shot matching call go before and after the kernels? create / set_regions (H2D of the regions +
tile build) / first run / second run / close, 1 000 images x 2 000 descriptors in separate pageable host arrays (as an SfM
pipeline holds them)."""
import json, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmvg_amd import matching, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
descs = synth.image_descriptors(n, n_desc=2000)
pairs = matching.exhaustive_pairs_array(n)
out = {"images": n}
for rep in range(2):
    t0 = time.perf_counter(); ctx = matching.MatchContext(0); t1 = time.perf_counter()
    ctx.set_regions(descs); t2 = time.perf_counter()
    ctx.run_stream(pairs, np.float32(0.64)); t3 = time.perf_counter()
    ctx.run_stream(pairs, np.float32(0.64)); t4 = time.perf_counter()
    ctx.close(); t5 = time.perf_counter()
    out[f"rep{rep}"] = {"create_ms": (t1 - t0) * 1e3, "set_regions_ms": (t2 - t1) * 1e3, "first_run_ms": (t3 - t2) * 1e3,
                        "second_run_ms": (t4 - t3) * 1e3, "close_ms": (t5 - t4) * 1e3}
print(json.dumps(out))
