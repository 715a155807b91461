"""(needs a library built with -DMVGX_FILTER_TIMING_VARIANTS: the forms 1..7 of the filter kernel with parts compiled out return wrong
lists and are not part of the shipped library) Where the time of l2_filter_kernel goes: the kernel re-timed with parts removed (results invalid, timing experiments only;
option "debug_filter" bit 0 = no epilogue, bit 1 = no per-tile LDS fragment loads, bit 2 = no per-window wait / barrier / staging).
400 images x 2000 descriptors, 79 800 pairs in one batch per run, overlap off (isolated kernel time). One JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmvg_amd import matching, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
descs = synth.image_descriptors(n, n_desc=2000, seed=0xC0FFEE00)
pairs = matching.exhaustive_pairs_array(n)
r2 = np.float32(0.8) * np.float32(0.8)
out = {"images": n, "image_pairs": int(len(pairs))}
ctx = matching.MatchContext(0)
ctx.set_option("profile", 1); ctx.set_option("overlap", 0); ctx.set_option("batch_pairs", 32768); ctx.set_option("keep_host_results", 0)
ctx.set_regions(descs)
names = {0: "full kernel", 1: "no epilogue", 2: "no fragment loads", 3: "no epilogue, no fragment loads", 4: "no window barrier/staging",
         5: "no epilogue, no barrier", 6: "no fragment loads, no barrier", 7: "MFMA stream only"}
for dbg in (0, 1, 2, 4, 3, 7, 0):
    ctx.set_option("debug_filter", dbg)
    ctx.run(pairs, r2, fetch=False)
    st, _, _ = ctx.run(pairs, r2, fetch=False)
    tops = float(st.n_desc_pairs) * 256 / (st.kernel_ms * 1e-3) / 1e12
    out.setdefault("runs", []).append({"debug_filter": dbg, "what": names[dbg], "kernel_ms": st.kernel_ms, "launches": int(st.n_kernel_launches),
                                       "tops": tops, "frac_of_5000": tops / 5000.0})
ctx.close()
print(json.dumps(out))
