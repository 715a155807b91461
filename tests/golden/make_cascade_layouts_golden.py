"""Generates tests/golden/cascade_layouts.npz: the reference's CascadeHasher (oracle/_ref, ref_cascade_match_pair_u8) on one pair
with several bucket layouts - few bits per bucket put hundreds of candidates into a bucket, where the top-ten selection, the
repeat test across groups and the (distance, id) tie order do real work - with duplicated database rows and a query equal to a
database row; per layout the hash outputs and the match list before de-duplication at ratio 0.8 and 1.3.
  python tests/golden/make_cascade_layouts_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from openmvg_amd import synth  # noqa: E402
from tests import _oracle  # noqa: E402

LAYOUTS = [(6, 10), (6, 2), (3, 1), (8, 4), (1, 3), (6, 6), (8, 16)]


def main():
    dI, dJ = synth.image_descriptors(2, n_desc=400, seed=9)
    dI = dI.copy(); dJ = dJ[:333].copy()
    dI[20] = dI[10]; dI[21] = dI[10]          # equal database rows: equal distances, the lower id wins
    dJ[5] = dI[10]                            # distance 0 to three rows
    dJ[7] = dI[300]
    out = {"descI": dI, "descJ": dJ, "layouts": np.array(LAYOUTS, np.int32)}
    for g, b in LAYOUTS:
        for ratio in (0.8, 1.3):
            m, hI, bI, hJ, bJ = _oracle.ref_cascade_match_pair(dI, dJ, ratio, g, b)
            out[f"g{g}b{b}/r{int(ratio * 10)}"] = m
            print(g, b, ratio, len(m))
        out[f"g{g}b{b}/hashI"] = hI; out[f"g{g}b{b}/bidsI"] = bI; out[f"g{g}b{b}/hashJ"] = hJ; out[f"g{g}b{b}/bidsJ"] = bJ
    np.savez_compressed(os.path.join(HERE, "cascade_layouts.npz"), **out)


if __name__ == "__main__":
    main()
