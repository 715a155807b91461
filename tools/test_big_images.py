"""Large feature counts per image (real SIFT gives 10-40 k): GPU lists vs the C restatement on a few pairs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmvg_amd import matching, synth
from tests import _oracle
for n in (9000, 30000, 70001):
    imgs = synth.image_descriptors(3, n_desc=n, seed=11)
    imgs[2] = imgs[2][: n // 3 + 1]
    pairs = np.array([[0, 1], [1, 0], [0, 2], [2, 1]], np.uint32)
    ctx = matching.MatchContext(0); ctx.set_regions(imgs)
    t = time.perf_counter(); st, off, ij = ctx.run(pairs, np.float32(0.64)); dt = time.perf_counter() - t
    ctx.close()
    o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pairs, 0.8)
    print(n, "equal" if (np.array_equal(off, o_off) and np.array_equal(ij, o_ij)) else "DIFFERENT", int(off[-1]), int(o_off[-1]), f"{dt:.3f}s", flush=True)
