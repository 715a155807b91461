"""End-to-end wall time of Cascade_Hashing_Matcher_Regions::Match (CASCADE_HASHING_L2, container included): the MI355X replacement
TU (hashing stage on the host threads, matching stage on the device) vs the reference TU, same caller code
(oracle/ref_shim_match.cpp::ref_cascade_matcher_regions_match_u8). One JSON line per set size."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmvg_amd import matching, synth
from tests import _oracle

for n, with_ref in ((60, True), (300, True), (1000, "--ref-1000" in sys.argv)):
    descs = synth.image_descriptors(n, n_desc=2000, seed=0xC0FFEE00)
    rng = np.random.default_rng(3)
    xy = [(rng.random((len(d), 2)) * 4000).astype(np.float32) for d in descs]
    pairs = matching.exhaustive_pairs_array(n)
    rec = {"images": n, "image_pairs": int(len(pairs))}
    for rep in range(2):
        t0 = time.perf_counter(); got = _oracle.ref_cascade_matcher_regions_match(descs, xy, pairs, 0.8, lib=_oracle.adapter()); dt = time.perf_counter() - t0
        rec.setdefault("replacement_s", []).append(round(dt, 3))
    rec["matches"] = int(sum(len(v) for v in got.values())); rec["pairs_with_matches"] = len(got)
    if with_ref and _oracle.have_ref_match():
        t0 = time.perf_counter(); ref = _oracle.ref_cascade_matcher_regions_match(descs, xy, pairs, 0.8); dt = time.perf_counter() - t0
        rec["reference_s"] = round(dt, 3)
        rec["identical"] = bool(ref.keys() == got.keys() and all(np.array_equal(ref[k], got[k]) for k in ref))
        rec["speedup"] = round(dt / min(rec["replacement_s"]), 1)
    print(json.dumps(rec), flush=True)
