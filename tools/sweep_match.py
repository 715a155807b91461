"""Within-process A/B of the matching kernel variants (interleaved rounds, median/min reported)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmvg_amd import matching, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=300)
    ap.add_argument("--desc", type=int, default=2000)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--variants", type=str, default="1,2,3")
    ap.add_argument("--out", type=str, default="")
    a = ap.parse_args()
    descs = synth.image_descriptors(a.images, n_desc=a.desc)
    pairs = matching.exhaustive_pairs_array(a.images)
    ctxs = {}
    for v in [int(x) for x in a.variants.split(",")]:
        c = matching.MatchContext(0)
        if v >= 40:
            c.set_option("variant", 4)
            c.set_option("stage", v - 40)
        else:
            c.set_option("variant", v)
        c.set_option("profile", 2)   # 2: also count the filter's candidates
        c.set_regions(descs)
        c.run(pairs[:1000], 0.64, fetch=False)
        ctxs[v] = c
    res = {v: [] for v in ctxs}
    for _ in range(a.rounds):
        for v, c in ctxs.items():
            t0 = time.perf_counter()
            st, _, _ = c.run(pairs, 0.64, fetch=False)
            dt = time.perf_counter() - t0
            res[v].append((st.n_desc_pairs / (st.kernel_ms * 1e-3), st.n_desc_pairs / dt, st.kernel_ms, dt * 1e3))
    rows = []
    for v, r in res.items():
        r = np.array(r)
        rows.append({"variant": v, "kernel_dpps_median": float(np.median(r[:, 0])), "kernel_dpps_best": float(r[:, 0].max()),
                     "e2e_dpps_median": float(np.median(r[:, 1])), "kernel_ms_median": float(np.median(r[:, 2])),
                     "e2e_ms_median": float(np.median(r[:, 3])),
                     "mfma_frac_of_5PF": float(np.median(r[:, 0]) * 256 / 5e15), "candidates_last_run": int(st.kernel_vgprs), "queries_last_run": int(st.n_pairs) * a.desc})
        print(json.dumps(rows[-1]), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
