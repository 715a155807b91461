"""Generates tests/golden/cascade_hashing.npz: inputs and outputs of the reference's CASCADE_HASHING_L2 path for the parity tests
that run without /root/reference (GPU box): descriptors, feature positions, the hashing stage's per-descriptor outputs (hash code,
bucket ids: single-precision Eigen products of the reference's CascadeHasher, oracle/_ref) and the final match lists of
Cascade_Hashing_Matcher_Regions::Match at ratio 0.8 and 0.6 - on a synthetic 5-image set with an empty image and repeated feature
positions, and on the real SIFT regions of tests/golden/sceaux_sift.npz.   python tests/golden/make_cascade_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from openmvg_amd import matching, synth  # noqa: E402
from tests import _oracle  # noqa: E402


def case(out, tag, descs, xy, pairs):
    hs, bs = _oracle.ref_cascade_hash(descs)
    out[f"{tag}/n_images"] = np.array(len(descs))
    out[f"{tag}/pairs"] = pairs
    for k in range(len(descs)):
        out[f"{tag}/desc{k}"] = descs[k]; out[f"{tag}/xy{k}"] = xy[k]; out[f"{tag}/hash{k}"] = hs[k]; out[f"{tag}/bids{k}"] = bs[k]
    for ratio in (0.8, 0.6):
        ref = _oracle.ref_cascade_matcher_regions_match(descs, xy, pairs, ratio)
        out[f"{tag}/r{int(ratio * 100)}/keys"] = np.array(sorted(ref), np.uint32).reshape(-1, 2)
        for (i, j), m in ref.items():
            out[f"{tag}/r{int(ratio * 100)}/{i}_{j}"] = m
        print(tag, ratio, {k: len(v) for k, v in ref.items()})


def main():
    out = {}
    descs = synth.image_descriptors(5, n_desc=700, seed=41)
    descs[2] = descs[2][:0]
    descs[4] = descs[4][:3]
    rng = np.random.default_rng(7)
    xy = [(rng.random((len(d), 2)) * 3000).astype(np.float32) for d in descs]            # distinct coordinates
    for a in xy:
        assert len(np.unique(a[:, 0])) == len(a) and len(np.unique(a[:, 1])) == len(a)
    case(out, "synthetic", descs, xy, matching.exhaustive_pairs_array(5))
    grid = [np.round(a / 100).astype(np.float32) for a in xy]                             # 30 x 30 grid: repeated positions, the
    case(out, "synthetic_grid", descs, grid, matching.exhaustive_pairs_array(5))          # coordinate de-duplication removes matches
    z = np.load(os.path.join(HERE, "sceaux_sift.npz"))
    case(out, "sceaux", [z["desc0"], z["desc1"]], [z["feat0"][:, :2].copy(), z["feat1"][:, :2].copy()], np.array([[0, 1]], np.uint32))
    np.savez_compressed(os.path.join(HERE, "cascade_hashing.npz"), **out)


if __name__ == "__main__":
    main()
