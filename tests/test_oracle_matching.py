"""CPU tests: pin oracle/match_oracle.c against (a) the golden values of the reference's own unit tests and
(b) the reference itself compiled in place (oracle/_ref/libref_match.so), on random and adversarial input."""
import ctypes as C

import numpy as np
import pytest

from tests import _oracle


def _l2(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return _oracle.port().oracle_l2_u8(a.ctypes.data, b.ctypes.data, a.size)


def _search(db, q, NN, lib=None, fn="oracle_search_neighbours_u8"):
    lib = lib or _oracle.port()
    db = np.ascontiguousarray(db, np.uint8); q = np.ascontiguousarray(q, np.uint8)
    nI, dim = (db.shape if db.size else (0, q.shape[1] if q.ndim == 2 else 1))
    nJ = q.shape[0] if q.size else 0
    idx = np.full(max(nJ * NN, 1), -1, np.int32); dist = np.full(max(nJ * NN, 1), -1, np.int32)
    ok = getattr(lib, fn)(db.ctypes.data if db.size else None, nI, q.ctypes.data if q.size else None, nJ, dim, NN,
                          idx.ctypes.data, dist.ctypes.data)
    return ok, idx[: nJ * NN].reshape(nJ, NN), dist[: nJ * NN].reshape(nJ, NN)


# ---- golden vectors from the reference's unit tests -------------------------------------------------

def test_metric_l2_golden_168():
    """metric_test.cpp:25-39: L2 of {0..7} vs {7..0} == 168 for every scalar type."""
    a = np.arange(8); b = a[::-1].copy()
    assert _l2(a, b) == 168
    ai = a.astype(np.int32); bi = b.astype(np.int32)
    assert _oracle.port().oracle_l2_i32(ai.ctypes.data, bi.ctypes.data, 8) == 168
    af = a.astype(np.float32); bf = b.astype(np.float32)
    assert _oracle.port().oracle_l2_f32(af.ctypes.data, bf.ctypes.data, 8) == 168.0


def test_metric_l2_dim128_matches_integer_squared_norm():
    """metric_test.cpp:132-147 (L2DIM128): uint8 128-D L2 == (a.cast<int>() - b.cast<int>()).squaredNorm()."""
    rng = np.random.default_rng(5)
    for _ in range(200):
        a = rng.integers(0, 256, 128, dtype=np.uint8); b = rng.integers(0, 256, 128, dtype=np.uint8)
        gt = int(((a.astype(np.int64) - b.astype(np.int64)) ** 2).sum())
        assert _l2(a, b) == gt
    a = np.zeros(128, np.uint8); b = np.full(128, 255, np.uint8)
    assert _l2(a, b) == 128 * 255 * 255  # the maximum, 8 323 200 < 2^24


def test_bruteforce_simple_dim1():
    """matching_test.cpp:26-39: array {0,1,2,3,4}, query {2} -> index 2, distance 0."""
    ok, idx, dist = _search(np.array([[0], [1], [2], [3], [4]]), np.array([[2]]), 1)
    assert ok and idx[0, 0] == 2 and dist[0, 0] == 0


def test_bruteforce_nn5_exact_order():
    """matching_test.cpp:41-69: array {0,1,2,5,6}, query {2}, 5-NN -> indices 2,1,0,3,4 and squared distances."""
    ok, idx, dist = _search(np.array([[0], [1], [2], [5], [6]]), np.array([[2]]), 5)
    assert ok
    assert idx[0].tolist() == [2, 1, 0, 3, 4]
    assert dist[0].tolist() == [0, 1, 4, 9, 16]


def test_bruteforce_simple_dim4():
    """matching_test.cpp:71-87."""
    ok, idx, dist = _search(np.arange(12).reshape(3, 4), np.array([[4, 5, 6, 7]]), 1)
    assert ok and idx[0, 0] == 1 and dist[0, 0] == 0


def test_bruteforce_empty_arrays():
    """matching_test.cpp:155-163 + matcher_brute_force.hpp:108-113: no database / NN > rows / no query -> false."""
    ok, _, _ = _search(np.zeros((0, 4)), np.array([[1, 2, 3, 4]]), 1)
    assert not ok
    ok, _, _ = _search(np.arange(4).reshape(1, 4), np.array([[1, 2, 3, 4]]), 2)  # NN=2 > 1 row
    assert not ok
    ok, _, _ = _search(np.arange(8).reshape(2, 4), np.zeros((0, 4)), 2)
    assert not ok


# ---- the restatement against the reference compiled in place ----------------------------------------

needs_ref = pytest.mark.skipif(not _oracle.have_ref_match(), reason="oracle/_ref/libref_match.so not built")


@needs_ref
def test_ref_is_the_avx2_build_and_agrees_on_metric():
    R = _oracle.ref_match()
    assert R.ref_uses_avx2() == 1
    rng = np.random.default_rng(11)
    for n in (8, 64, 128):
        for _ in range(50):
            a = rng.integers(0, 256, n, dtype=np.uint8); b = rng.integers(0, 256, n, dtype=np.uint8)
            assert R.ref_l2_u8(a.ctypes.data, b.ctypes.data, n) == _l2(a, b)
    a = np.arange(8, dtype=np.uint8); b = a[::-1].copy()
    assert R.ref_l2_u8(a.ctypes.data, b.ctypes.data, 8) == 168


@needs_ref
def test_port_2nn_equals_reference_2nn_distances():
    """Distances of the two nearest neighbours are identical; indices are identical wherever d0 < d1."""
    rng = np.random.default_rng(3)
    db = rng.integers(0, 256, (257, 128), dtype=np.uint8)
    q = rng.integers(0, 256, (131, 128), dtype=np.uint8)
    q[5] = db[17]; q[6] = db[200]
    okp, ip, dp = _search(db, q, 2)
    okr, ir, dr = _search(db, q, 2, lib=_oracle.ref_match(), fn="ref_search_neighbours_u8")
    assert okp and okr
    assert np.array_equal(dp, dr)
    strict = dp[:, 0] < dp[:, 1]
    assert np.array_equal(ip[strict, 0], ir[strict, 0])
    assert ip[5, 0] == 17 and dp[5, 0] == 0


def _adversarial_set():
    rng = np.random.default_rng(99)
    base = rng.integers(0, 256, (70, 128), dtype=np.uint8)
    imgs = []
    imgs.append(base[:40].copy())                                   # 0: plain
    a = base[20:60].copy(); a[3] = a[4]                             # 1: overlaps 0, holds an exact duplicate row
    imgs.append(a)
    z = base[10:50].copy(); z[0] = 0; z[1] = 0; z[2] = 255           # 2: all-zero rows (duplicates) and a saturated row
    imgs.append(z)
    imgs.append(np.zeros((0, 128), np.uint8))                       # 3: empty image
    imgs.append(base[:1].copy())                                    # 4: one descriptor  (NN=2 > rows as database)
    imgs.append(base[:2].copy())                                    # 5: two descriptors
    n = base[:33].copy().astype(np.int16) + rng.integers(-2, 3, (33, 128))
    imgs.append(np.clip(n, 0, 255).astype(np.uint8))                # 6: noisy copy of 0 -> many accepted matches
    imgs.append(np.repeat(base[60:61], 35, axis=0))                 # 7: 35 identical rows: every query has d0 == d1
    return imgs


@needs_ref
@pytest.mark.parametrize("ratio", [0.8, 0.6, 1.0])
def test_port_matcher_regions_equals_reference_on_adversarial_set(ratio):
    from openmvg_amd.matching import exhaustive_pairs_array
    imgs = _adversarial_set()
    n = len(imgs)
    pairs = np.concatenate([exhaustive_pairs_array(n), exhaustive_pairs_array(n)[:, ::-1]])  # both orientations
    offsets, ij = _oracle.port_matcher_regions_match(imgs, pairs, ratio)
    got = _oracle.offsets_to_dict(pairs, offsets, ij)
    ref = _oracle.ref_matcher_regions_match(imgs, pairs, ratio)
    assert set(got) == set(ref)
    for k in ref:
        assert np.array_equal(got[k], ref[k]), k
    assert (0, 6) in ref and len(ref[(0, 6)]) > 20           # the planted matches are found
    assert all(3 not in k for k in ref)                       # empty image never reported
    assert all(k[0] != 4 for k in ref)                        # 1-row database never reported


@needs_ref
def test_port_matcher_regions_equals_reference_on_rootsift_like_set():
    from openmvg_amd import synth
    from openmvg_amd.matching import exhaustive_pairs_array
    imgs = synth.image_descriptors(6, n_desc=300, seed=21)
    pairs = exhaustive_pairs_array(6)
    offsets, ij = _oracle.port_matcher_regions_match(imgs, pairs, 0.8)
    got = _oracle.offsets_to_dict(pairs, offsets, ij)
    ref = _oracle.ref_matcher_regions_match(imgs, pairs, 0.8)
    assert set(got) == set(ref) and len(ref) > 0
    for k in ref:
        assert np.array_equal(got[k], ref[k])
