"""Guided matching on the device: n_pairs image pairs x n features per image (synthetic two-view geometry with clutter), timing of
mvgx_guided_match_u8. Usage: guided_run.py [n_pairs] [features] [kind 0|1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmvg_amd import geofilter
from tests.test_guided_matching import _pair
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
kind = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rng = np.random.default_rng(11)
base = [_pair(rng, n // 2, n - n // 2, n - n // 2, kind) for _ in range(8)]   # eight distinct pairs, cycled
feats, descs, pairs, models = [], [], [], []
for xi, di, xj, dj, M in base:
    feats += [xi, xj]; descs += [di, dj]
for p in range(n_pairs):
    b = p % 8
    pairs.append((2 * b, 2 * b + 1)); models.append(base[b][4])
prec = np.full(n_pairs, 4.0)
geofilter.guided_matching(feats, descs, pairs[:64], models[:64], prec[:64], 0.8, kind)   # warm-up
for rep in range(3):
    t0 = time.perf_counter()
    res, st = geofilter.guided_matching(feats, descs, pairs, models, prec, 0.8, kind)
    dt = time.perf_counter() - t0
    print(f"kind {kind}: {n_pairs} pairs x {n} x {n}: kernel {st.kernel_ms:.2f} ms, call {st.total_ms:.2f} ms, wall {dt * 1e3:.1f} ms; {st.n_geometric_tests / (st.kernel_ms * 1e-3):.4g} geometric tests/s, "
          f"{n_pairs / (st.total_ms * 1e-3):.4g} pairs/s whole call; passed {st.n_geometric_passed} ({st.n_geometric_passed / st.n_geometric_tests:.2e}), descriptor stages {st.n_descriptor_stages}, matches {st.n_matches}", flush=True)
