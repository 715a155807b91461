"""Generates tests/golden/matching_golden.npz by running the REFERENCE's own Matcher_Regions (compiled in place into
oracle/_ref/libref_match.so) on small fixed inputs. Run in the build container (needs /root/reference):

    python tests/golden/make_matching_golden.py

The .npz travels to the GPU box (no /root/reference there); tests compare the HIP path and the C oracle against it.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from openmvg_amd import synth  # noqa: E402
from openmvg_amd.matching import exhaustive_pairs_array  # noqa: E402
from tests import _oracle  # noqa: E402
from tests.test_oracle_matching import _adversarial_set  # noqa: E402


def pack(prefix, imgs, pairs, ratio, out):
    ref = _oracle.ref_matcher_regions_match(imgs, pairs, ratio)
    out[f"{prefix}_n"] = np.array([len(i) for i in imgs], np.uint32)
    out[f"{prefix}_desc"] = np.concatenate([i.reshape(-1, 128) for i in imgs]).astype(np.uint8)
    out[f"{prefix}_pairs"] = pairs
    out[f"{prefix}_ratio"] = np.float32(ratio)
    keys = sorted(ref)
    out[f"{prefix}_ref_pairs"] = np.array(keys, np.uint32).reshape(-1, 2)
    out[f"{prefix}_ref_counts"] = np.array([len(ref[k]) for k in keys], np.uint32)
    out[f"{prefix}_ref_ij"] = (np.concatenate([ref[k] for k in keys]) if keys else np.zeros((0, 2))).astype(np.uint32)


def main():
    out = {}
    adv = _adversarial_set()
    n = len(adv)
    both = np.concatenate([exhaustive_pairs_array(n), exhaustive_pairs_array(n)[:, ::-1]])
    pack("adv08", adv, both, 0.8, out)
    pack("adv10", adv, both, 1.0, out)
    sift = synth.image_descriptors(6, n_desc=300, seed=21)
    pack("sift08", sift, exhaustive_pairs_array(6), 0.8, out)
    rag = synth.random_descriptors(7, [1, 2, 31, 33, 257, 300, 64], seed=4)
    rag[5][:257] = np.clip(rag[4].astype(np.int16) + np.random.default_rng(8).integers(-6, 7, (257, 128)), 0, 255).astype(np.uint8)
    rag_pairs = np.concatenate([exhaustive_pairs_array(7), exhaustive_pairs_array(7)[:, ::-1]])
    pack("ragged06", rag, rag_pairs, 0.6, out)
    path = os.path.join(ROOT, "tests", "golden", "matching_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if k.endswith("ref_pairs")})


if __name__ == "__main__":
    main()
