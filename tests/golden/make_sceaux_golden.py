"""Generates tests/golden/sceaux_sift.npz: REAL descriptors for the matching parity tests (BASELINE configs[0]).

The two JPGs that ship with the reference (openMVG_Samples/imageData/SceauxCastle/100_7101.jpg, 100_7102.jpg) are described with
the reference's own SIFT (features/sift/SIFT_Anatomy_Image_Describer.hpp, default parameters, RootSIFT) through
oracle/_ref/libref_features.so, then matched in both directions at ratio 0.8 (main_ComputeMatches' default) and 0.6 with the
reference's own Matcher_Regions (oracle/_ref/libref_match.so). Runs in the build container only (needs /root/reference);
the fixture it writes travels with the repository.  python tests/golden/make_sceaux_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests import _oracle  # noqa: E402

IMAGES = ["100_7101.jpg", "100_7102.jpg"]
DIR = "/root/reference/src/openMVG_Samples/imageData/SceauxCastle"


def describe(lib, path):
    n = lib.ref_sift_describe_jpeg(path.encode())
    assert n > 0, (path, n)
    desc = np.zeros((n, 128), np.uint8)
    feat = np.zeros((n, 4), np.float32)
    lib.ref_sift_copy(desc.ctypes.data_as(C.c_void_p), feat.ctypes.data_as(C.c_void_p))
    w, h = C.c_int(), C.c_int()
    lib.ref_sift_image_size(C.byref(w), C.byref(h))
    return desc, feat, (w.value, h.value)


def main():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_features.so"))
    lib.ref_sift_describe_jpeg.argtypes = [C.c_char_p]
    out = {"images": np.array(IMAGES)}
    descs = []
    for k, name in enumerate(IMAGES):
        d, f, size = describe(lib, os.path.join(DIR, name))
        print(name, size, len(d), "regions")
        out[f"desc{k}"] = d; out[f"feat{k}"] = f; out[f"size{k}"] = np.array(size, np.int32)
        descs.append(d)
    pairs = np.array([[0, 1], [1, 0]], np.uint32)
    for ratio in (0.8, 0.6):
        ref = _oracle.ref_matcher_regions_match(descs, pairs, ratio)
        for (i, j), m in ref.items():
            out[f"matches_r{int(ratio * 100)}_{i}_{j}"] = m
            print(f"ratio {ratio}: ({i}, {j}) -> {len(m)} putative matches")
    np.savez_compressed(os.path.join(HERE, "sceaux_sift.npz"), **out)


if __name__ == "__main__":
    main()
