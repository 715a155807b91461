#!/usr/bin/env python
"""Reference against reference: how far do two builds of the SAME openMVG sources differ on the geometric filter?

oracle/_ref/libref_geofilter.so      = the reference's kernel + ACRANSAC, g++ -O3 (portable x86-64: SSE2, no contraction)
oracle/_ref/libref_geofilter_fma.so  = the same sources and the same shim, g++ -O3 -mavx2 -mfma (`make -C oracle ref_geofilter_fma`;
                                       what -march=native does on any current x86 host: Eigen's AVX kernels + fused a*b+c)

Both run on the bench sample (the first N pairs of bench_geofilter.py's set, same seed) and on the golden fixtures; the result is
the share of pairs on which the two builds end with different inlier sets. VERDICT r3 asked for this number: the device-vs-reference
checks are bounded by it instead of a flat 1 - 2 %.  CPU only; writes one JSON record (committed as
profiles/round4_geofilter_reference_vs_reference.json).
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmvg_amd import synth            # noqa: E402
from tests import _oracle                # noqa: E402
from tests import _geofilter_cases as gc  # noqa: E402


def differing(start, a, b):
    start = np.asarray(start, np.int64)
    out = []
    for p in range(len(start) - 1):
        lo, hi = start[p], start[p + 1]
        if bool(a["ok"][p]) != bool(b["ok"][p]) or not np.array_equal(a["mask"][lo:hi], b["mask"][lo:hi]):
            out.append(p)
    return out


def run(model, n_pairs, n):
    """the first n_pairs pairs of bench_geofilter.py's set (100 000 pairs for F, 20 000 for H: the generators are not prefix-stable)"""
    fma = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_geofilter_fma.so"))
    if model == "h":
        tv = synth.two_view_homography_matches(max(n_pairs, 20000), seed=0x6E0F, n_min=n, n_max=n, tiny_frac=0.0)
    elif model == "e":
        tv = synth.two_view_matches_bulk(max(n_pairs, 20000), n=n, seed=0x6E0F)
    else:
        tv = synth.two_view_matches_bulk(max(n_pairs, 100000), n=n, seed=0x6E0F)
    tv = dict(xI=tv["xI"][:n * n_pairs], xJ=tv["xJ"][:n * n_pairs], start=tv["start"][:n_pairs + 1], wh=tv["wh"][:n_pairs])
    if model == "e":
        K = synth.two_view_calibration(tv)
        a = _oracle.ref_geofilter_e(tv, K)
        b = _oracle._geofilter_call_e(fma.ref_geofilter_e_acransac, tv, K, False, 4.0, 2048, 0)
    elif model == "h":
        a = _oracle.ref_geofilter_h(tv)
        b = _oracle._geofilter_call(fma.ref_geofilter_h_acransac, tv, 4.0, 2048, 0)
    else:
        a = _oracle.ref_geofilter(tv)
        b = _oracle._geofilter_call(fma.ref_geofilter_f_acransac, tv, 4.0, 2048, 0)
    d = differing(tv["start"], a, b)
    same = np.ones(n_pairs, bool); same[d] = False
    both = same & a["ok"]
    dF = float(np.abs(gc.normalised(a["F"][both]) - gc.normalised(b["F"][both])).max()) if both.any() else 0.0
    nfa_rel = float(np.max(np.abs(a["nfa"][both] - b["nfa"][both]) / np.maximum(1.0, np.abs(a["nfa"][both])))) if both.any() else 0.0
    # on the differing pairs: which side found the more meaningful model (lower NFA)?
    detail = [dict(pair=int(p), ok=[bool(a["ok"][p]), bool(b["ok"][p])], nfa=[float(a["nfa"][p]), float(b["nfa"][p])],
                   inliers=[int(a["mask"][tv["start"][p]:tv["start"][p + 1]].sum()), int(b["mask"][tv["start"][p]:tv["start"][p + 1]].sum())])
              for p in d[:40]]
    return {"model": model, "pairs": n_pairs, "matches_per_pair": n, "pairs_ok_O3": int(a["ok"].sum()), "pairs_ok_O3_avx2_fma": int(b["ok"].sum()),
            "pairs_differing": len(d), "share": len(d) / n_pairs, "max_abs_dF_on_equal_sets": dF, "max_rel_dNFA_on_equal_sets": nfa_rel,
            "seconds": [a["seconds"], b["seconds"]], "first_differing": detail}


if __name__ == "__main__":
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
    n_h = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    rec = {"what": "compiled reference (-O3) vs compiled reference (-O3 -mavx2 -mfma) on the bench samples of bench_geofilter.py",
           "sources": "openMVG/robust_estimation/robust_estimator_ACRansac.hpp, multiview/solver_fundamental_kernel.cpp, "
                      "multiview/solver_homography_kernel.cpp through oracle/ref_shim_geofilter.cpp (both builds)",
           "f": run("f", n_pairs, 250), "h": run("h", n_h, 250), "e": run("e", n_h, 250)}
    print(json.dumps(rec, indent=1))
