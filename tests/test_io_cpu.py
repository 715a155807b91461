"""CPU tests of the file formats either side of the matching path (openmvg_amd/io.py) against the
reference's own readers / writers (oracle/_ref/libref_io.so, compiled from the reference tree by oracle/Makefile) and
against small files written by the reference and committed under tests/golden/io/."""
import ctypes as C
import os

import numpy as np
import pytest

from openmvg_amd import io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_IO_SO = os.path.join(ROOT, "oracle", "_ref", "libref_io.so")
GOLD = os.path.join(ROOT, "tests", "golden", "io")
needs_ref = pytest.mark.skipif(not os.path.exists(REF_IO_SO), reason="oracle/_ref/libref_io.so not built")


def _ref():
    L = C.CDLL(REF_IO_SO)
    L.ref_io_save_desc.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64]
    L.ref_io_load_desc.restype = C.c_int64
    L.ref_io_load_desc.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64]
    L.ref_io_save_feat.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64]
    L.ref_io_load_feat.restype = C.c_int64
    L.ref_io_load_feat.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64]
    L.ref_io_save_matches_txt.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.ref_io_load_matches_txt.restype = C.c_int64
    L.ref_io_load_matches_txt.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    return L


def _sample():
    rng = np.random.default_rng(3)
    desc = rng.integers(0, 256, (37, 128), dtype=np.uint8)
    feats = np.stack([rng.uniform(0, 4000, 37), rng.uniform(0, 3000, 37), rng.uniform(0.5, 40, 37), rng.uniform(-3.2, 3.2, 37)], 1).astype(np.float32)
    pairs = np.array([[0, 3], [0, 1], [2, 5], [1, 2]], np.uint32)
    offsets = np.array([0, 3, 3, 8, 9], np.uint64)            # pair (0,1) has no matches: it must not appear in the file
    ij = rng.integers(0, 2000, (9, 2)).astype(np.uint32)
    return desc, feats, pairs, offsets, ij


def _matches_dict(pairs, offsets, ij):
    return {(int(a), int(b)): ij[int(offsets[k]):int(offsets[k + 1])] for k, (a, b) in enumerate(pairs) if offsets[k + 1] > offsets[k]}


@needs_ref
def test_round_trips_through_the_reference_code(tmp_path):
    L = _ref()
    desc, feats, pairs, offsets, ij = _sample()
    # .desc: ours -> reference reader, reference writer -> ours, byte-identical files
    a, b = str(tmp_path / "a.desc"), str(tmp_path / "b.desc")
    io.save_desc_bin(a, desc)
    n = L.ref_io_load_desc(a.encode(), None, 0)
    got = np.zeros((n, 128), np.uint8)
    assert n == len(desc) and L.ref_io_load_desc(a.encode(), got.ctypes.data, n) == n and np.array_equal(got, desc)
    assert L.ref_io_save_desc(b.encode(), desc.ctypes.data, len(desc)) == 0
    assert open(a, "rb").read() == open(b, "rb").read() and np.array_equal(io.load_desc_bin(b), desc)
    # .feat
    a, b = str(tmp_path / "a.feat"), str(tmp_path / "b.feat")
    io.save_feat(a, feats)
    got = np.zeros((len(feats), 4), np.float32)
    assert L.ref_io_load_feat(a.encode(), got.ctypes.data, len(feats)) == len(feats)
    assert L.ref_io_save_feat(b.encode(), feats.ctypes.data, len(feats)) == 0
    assert open(a).read() == open(b).read()                                   # same "%g"-style text as operator<<
    assert np.allclose(got, io.load_feat(b), rtol=0, atol=0) and np.allclose(got, feats, rtol=1e-5)
    # matches.txt
    a, b = str(tmp_path / "a.txt"), str(tmp_path / "b.txt")
    io.save_matches_txt(a, pairs, offsets, ij)
    assert L.ref_io_save_matches_txt(b.encode(), pairs.ctypes.data, len(pairs), offsets.ctypes.data, ij.ctypes.data) == 0
    assert open(a).read() == open(b).read()
    rows = np.zeros((len(ij), 4), np.uint32); nm = C.c_uint64()
    assert L.ref_io_load_matches_txt(a.encode(), rows.ctypes.data, len(ij), C.byref(nm)) == 3 and nm.value == len(ij)
    want = _matches_dict(pairs, offsets, ij)
    got = io.load_matches_txt(b)
    assert got.keys() == want.keys() and all(np.array_equal(got[k], want[k]) for k in want)
    assert sorted({(int(r[0]), int(r[1])) for r in rows}) == sorted(want)


def test_files_written_by_the_reference():
    """tests/golden/io/*: written by the reference code (make_io_golden.py), read here without it"""
    desc, feats, pairs, offsets, ij = _sample()
    assert np.array_equal(io.load_desc_bin(os.path.join(GOLD, "sample.desc")), desc)
    assert np.allclose(io.load_feat(os.path.join(GOLD, "sample.feat")), feats, rtol=1e-5)
    want = _matches_dict(pairs, offsets, ij)
    got = io.load_matches_txt(os.path.join(GOLD, "matches.putative.txt"))
    assert got.keys() == want.keys() and all(np.array_equal(got[k], want[k]) for k in want)


def test_error_and_empty_cases(tmp_path):
    p = str(tmp_path / "e.desc")
    io.save_desc_bin(p, np.zeros((0, 128), np.uint8))
    assert io.load_desc_bin(p).shape == (0, 128)
    open(p, "wb").write(b"\x05\x00\x00\x00\x00\x00\x00\x00abc")     # header promises 5 descriptors
    with pytest.raises(ValueError):
        io.load_desc_bin(p)
    q = str(tmp_path / "e.txt")
    io.save_matches_txt(q, np.zeros((0, 2), np.uint32), np.zeros(1, np.uint64), np.zeros((0, 2), np.uint32))
    assert io.load_matches_txt(q) == {}


# ---- BAF export (sfm/sfm_data_io_baf.hpp:38-147) ----
REF_BA_SO = os.path.join(ROOT, "oracle", "_ref", "libref_ba.so")


def _baf_scene():
    from openmvg_amd import synth
    sc = synth.ba_scene(5, 23, track_len=3, model=3, n_intr_groups=2, seed=11, noise_px=0.3)
    sc["intr_model"] = np.array([3, 1], np.int32)      # radial-3 and pinhole in one file: 6 and 3 parameters per line
    return sc


def _baf_sections(path):
    """-> (header, sorted intrinsic lines, sorted view lines, sorted landmark lines); inside a section the reference's line
    order is the iteration order of std::unordered_map (types.hpp:67), so sections are compared as multisets."""
    lines = open(path).read().split("\n")
    assert lines[-1] == ""
    ni, nv, nl = (int(x) for x in lines[:3])
    body = lines[3:-1]
    assert len(body) == ni + nv + nl
    return lines[:3], sorted(body[:ni]), sorted(body[ni:ni + nv]), sorted(body[ni + nv:])


def _baf_landmark_canonical(line):
    """observations of one landmark sorted by id_pose (the reference emits them in hash-map order)."""
    t = line.split()
    n = int(t[3])
    obs = sorted(tuple(t[4 + 4 * k:8 + 4 * k]) for k in range(n))
    return tuple(t[:4]), tuple(obs)


def _write_ours(path, sc):
    io.save_baf(path, sc["poses"], sc["intrinsics"], sc["intr_model"], sc["points"], sc["obs_pose"], sc["obs_intr"],
                sc["obs_point"], sc["obs_xy"])


def _compare_baf(ours, ref):
    ho, io_, vo, lo = _baf_sections(ours)
    hr, ir, vr, lr = _baf_sections(ref)
    assert ho == hr and io_ == ir and vo == vr
    assert sorted(map(_baf_landmark_canonical, lo)) == sorted(map(_baf_landmark_canonical, lr))
    a = sorted(open(os.path.splitext(ours)[0] + "_imgList.txt").read().split("\n"))
    b = sorted(open(os.path.splitext(ref)[0] + "_imgList.txt").read().split("\n"))
    assert a == b


@pytest.mark.skipif(not os.path.exists(REF_BA_SO), reason="oracle/_ref/libref_ba.so not built")
def test_baf_export_matches_the_reference_writer(tmp_path):
    from tests import _oracle
    sc = _baf_scene()
    ours, ref = str(tmp_path / "ours.baf"), str(tmp_path / "ref.baf")
    _write_ours(ours, sc)
    assert _oracle.ref_save_baf(sc, ref) == 0
    _compare_baf(ours, ref)


def test_baf_export_against_committed_reference_file(tmp_path):
    sc = _baf_scene()
    ours = str(tmp_path / "ours.baf")
    _write_ours(ours, sc)
    _compare_baf(ours, os.path.join(GOLD, "scene.baf"))


# ---- file-level pipeline: <stem>.desc in, matches.putative.txt out ----
def test_match_directory_binary_descriptors_under_emulation(tmp_path):
    """io.match_directory on AKAZE-like binary descriptor files: the device code (emulated) behind the Matcher_Regions
    mirror, the written matches file read back and compared with the oracle - and with the reference's own
    Matcher_Regions(BRUTE_FORCE_HAMMING) when its build is present"""
    from openmvg_amd import matching, synth
    from tests import _emu, _oracle
    sizes = [120, 0, 300, 64, 257]
    imgs = synth.binary_descriptors(len(sizes), sizes, seed=41)
    stems = [f"img_{k:03d}" for k in range(len(sizes))]
    for stem, d in zip(stems, imgs):
        io.save_desc_bin(str(tmp_path / (stem + ".desc")), d)
    with _emu.emulated():
        got = io.match_directory(str(tmp_path), stems, ratio=0.8, kind="binary")
    back = io.load_matches_txt(str(tmp_path / "matches.putative.txt"))
    assert back.keys() == got.keys() and all(np.array_equal(back[k], got[k]) for k in got)
    pairs = matching.exhaustive_pairs_array(len(sizes))
    off, ij = _oracle.port_matcher_regions_match_hamming(imgs, pairs, 0.8)
    want = _oracle.offsets_to_dict(pairs, off, ij)
    assert len(want) >= 3 and back.keys() == want.keys() and all(np.array_equal(back[k], want[k]) for k in want)
    if _oracle.have_ref_match():
        ref = _oracle.ref_matcher_regions_match_binary64(imgs, pairs, 0.8)
        assert back.keys() == ref.keys() and all(np.array_equal(back[k], ref[k]) for k in ref)


def test_desc_files_of_the_other_region_types_round_trip(tmp_path):
    rng = np.random.default_rng(2)
    f = rng.standard_normal((17, 64)).astype(np.float32)
    b = rng.integers(0, 256, (9, 64), dtype=np.uint8)
    io.save_desc_bin(str(tmp_path / "f.desc"), f, np.float32); io.save_desc_bin(str(tmp_path / "b.desc"), b)
    assert np.array_equal(io.load_desc_bin(str(tmp_path / "f.desc"), 64, np.float32), f)
    assert np.array_equal(io.load_desc_bin(str(tmp_path / "b.desc"), 64), b)
    assert os.path.getsize(tmp_path / "f.desc") == 8 + 17 * 64 * 4
