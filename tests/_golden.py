import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "matching_golden.npz")
CASES = ("adv08", "adv10", "sift08", "ragged06")


def load_case(name):
    z = np.load(GOLDEN)
    n = z[f"{name}_n"]
    desc = z[f"{name}_desc"]
    offs = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    imgs = [desc[offs[k]:offs[k + 1]].copy() for k in range(len(n))]
    ref = {}
    cnt = z[f"{name}_ref_counts"]
    o = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    for k, (a, b) in enumerate(z[f"{name}_ref_pairs"]):
        ref[(int(a), int(b))] = z[f"{name}_ref_ij"][o[k]:o[k + 1]].copy()
    return imgs, z[f"{name}_pairs"].copy(), float(z[f"{name}_ratio"]), ref
