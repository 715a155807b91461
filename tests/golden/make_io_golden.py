"""Writes tests/golden/io/* with the REFERENCE's own writers (oracle/_ref/libref_io.so). Run in the build container."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_io_cpu import GOLD, _ref, _sample  # noqa: E402

L = _ref()
desc, feats, pairs, offsets, ij = _sample()
os.makedirs(GOLD, exist_ok=True)
assert L.ref_io_save_desc(os.path.join(GOLD, "sample.desc").encode(), desc.ctypes.data, len(desc)) == 0
assert L.ref_io_save_feat(os.path.join(GOLD, "sample.feat").encode(), feats.ctypes.data, len(feats)) == 0
assert L.ref_io_save_matches_txt(os.path.join(GOLD, "matches.putative.txt").encode(), pairs.ctypes.data, len(pairs), offsets.ctypes.data, ij.ctypes.data) == 0
from tests import _oracle  # noqa: E402
from tests.test_io_cpu import _baf_scene  # noqa: E402
assert _oracle.ref_save_baf(_baf_scene(), os.path.join(GOLD, "scene.baf")) == 0   # + scene_imgList.txt
print(os.listdir(GOLD))
