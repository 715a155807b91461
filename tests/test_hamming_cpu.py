"""CPU tests of the binary-descriptor path (BRUTE_FORCE_HAMMING, matching/regions_matcher.cpp:184-191):
the C restatement against the reference's own Matcher_Regions on AKAZE_Binary_Regions (compiled in place) and against
committed reference output; the device code of openmvg_amd/csrc/mvgx_bruteforce.hip under the HIP execution-model emulation
(tests/_emu.py) against the restatement."""
import os

import numpy as np
import pytest

from openmvg_amd import matching, synth
from tests import _emu, _oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hamming_golden.npz")


def golden_case():
    sizes = [0, 1, 2, 3, 63, 64, 65, 255, 257, 300]
    imgs = synth.binary_descriptors(len(sizes), sizes, seed=17)
    n = len(imgs)
    pairs = np.concatenate([matching.exhaustive_pairs_array(n), matching.exhaustive_pairs_array(n)[:, ::-1]])
    return imgs, pairs


def _as_dict(pairs, offsets, ij):
    return _oracle.offsets_to_dict(pairs, offsets, ij)


def _same(a, b):
    return a.keys() == b.keys() and all(np.array_equal(a[k], b[k]) for k in a)


@pytest.mark.parametrize("ratio", [0.8, 0.6, 1.0])
def test_restatement_equals_reference_and_golden(ratio):
    imgs, pairs = golden_case()
    off, ij = _oracle.port_matcher_regions_match_hamming(imgs, pairs, ratio)
    got = _as_dict(pairs, off, ij)
    assert sum(len(v) for v in got.values()) > 200
    g = np.load(GOLD)
    key = f"r{int(round(ratio * 100))}"
    assert np.array_equal(off, g[key + "_offsets"]) and np.array_equal(ij, g[key + "_ij"])
    if _oracle.have_ref_match():
        assert _same(got, _oracle.ref_matcher_regions_match_binary64(imgs, pairs, ratio))


@pytest.mark.skipif(not _oracle.have_ref_match(), reason="oracle/_ref/libref_match.so not built")
def test_hamming_metric_equals_reference():
    import ctypes as C
    rng = np.random.default_rng(4)
    L = _oracle.ref_match(); P = _oracle.port()
    L.ref_hamming_u8.restype = C.c_uint; L.ref_hamming_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    P.oracle_hamming_u8.restype = C.c_uint; P.oracle_hamming_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    for size in (1, 3, 4, 12, 32, 61, 64):   # the reference walks uint64 / uint32 / uint8 words depending on size
        a = rng.integers(0, 256, size, dtype=np.uint8); b = rng.integers(0, 256, size, dtype=np.uint8)
        want = int(np.unpackbits(a ^ b).sum())
        assert L.ref_hamming_u8(a.ctypes.data, b.ctypes.data, size) == want == P.oracle_hamming_u8(a.ctypes.data, b.ctypes.data, size)


def _run_emu(imgs, pairs, ratio, L=None, batch_pairs=None):
    with _emu.emulated():
        ctx = matching.HammingContext()
        if batch_pairs:
            ctx.set_option("batch_pairs", batch_pairs)
        ctx.set_regions(imgs, L)
        st, off, ij = ctx.run(pairs, ratio)
        ctx.close()
    return st, off, ij


@pytest.mark.parametrize("ratio,batch", [(0.8, None), (1.0, 7)])
def test_emulated_device_code_equals_restatement(ratio, batch):
    imgs, pairs = golden_case()
    o_off, o_ij = _oracle.port_matcher_regions_match_hamming(imgs, pairs, ratio)
    st, off, ij = _run_emu(imgs, pairs, ratio, 64, batch)
    assert np.array_equal(off, o_off) and np.array_equal(ij, o_ij)
    assert st.n_desc_pairs == sum(len(imgs[a]) * len(imgs[b]) for a, b in pairs if len(imgs[a]) >= 2 and len(imgs[b]))


@pytest.mark.parametrize("L", [32, 20, 61])
def test_emulated_other_descriptor_lengths(L):
    """32-byte descriptors take the 8-dword kernel; lengths that are no multiple of 4 are zero padded"""
    sizes = [40, 0, 300, 5]
    imgs = synth.binary_descriptors(len(sizes), sizes, n_bytes=L, seed=3, flip_bits=max(2, L // 2))
    pairs = np.array([(i, j) for i in range(4) for j in range(4) if i != j], np.uint32)
    o_off, o_ij = _oracle.port_matcher_regions_match_hamming(imgs, pairs, 0.9, L)
    _, off, ij = _run_emu(imgs, pairs, 0.9, L)
    assert int(o_off[-1]) > 5 and np.array_equal(off, o_off) and np.array_equal(ij, o_ij)


def test_emulated_duplicates_extremes_and_mirror():
    """all-zero vs all-one rows (d = 512), exact duplicates (d0 = d1 = 0: rejected for every ratio <= 1), and the
    Matcher_Regions mirror with BRUTE_FORCE_HAMMING"""
    rng = np.random.default_rng(8)
    a = np.zeros((70, 64), np.uint8); b = np.full((70, 64), 255, np.uint8)
    a[::3] = rng.integers(0, 256, (24, 64), dtype=np.uint8)
    b[::2] = a[::2]            # exact copies
    b[1] = b[3]                # a duplicate inside the database side
    c = rng.integers(0, 256, (70, 64), dtype=np.uint8); c[10] = a[12]; c[11] = a[12]
    imgs = [a, b, c]
    pairs = np.array([[0, 1], [1, 0], [0, 2], [2, 0], [1, 2], [2, 1]], np.uint32)
    for ratio in (1.0, 0.5):
        o_off, o_ij = _oracle.port_matcher_regions_match_hamming(imgs, pairs, ratio)
        _, off, ij = _run_emu(imgs, pairs, ratio, 64)
        assert np.array_equal(off, o_off) and np.array_equal(ij, o_ij)
    with _emu.emulated():
        prov = matching.Regions_Provider({10 + k: matching.Binary_Regions(d) for k, d in enumerate(imgs)})
        out = matching.PairWiseMatches()
        matching.Matcher_Regions(0.8, matching.EMatcherType.BRUTE_FORCE_HAMMING).Match(prov, [(10, 11), (11, 12), (10, 12)], out)
    o_off, o_ij = _oracle.port_matcher_regions_match_hamming(imgs, np.array([[0, 1], [0, 2], [1, 2]], np.uint32), 0.8)
    want = _oracle.offsets_to_dict(np.array([[10, 11], [10, 12], [11, 12]]), o_off, o_ij)
    assert _same(dict(out), want)


def test_emulated_error_behaviour():
    with _emu.emulated():
        ctx = matching.HammingContext()
        with pytest.raises(Exception):
            ctx.set_regions([np.zeros((3, 65), np.uint8)], 65)       # longer than the device path supports
        ctx.set_regions([np.zeros((3, 64), np.uint8)] * 2, 64)
        with pytest.raises(Exception):
            ctx.run(np.array([[0, 1]], np.uint32), 1.5)               # ratio > 1: tie order of libstdc++
        with pytest.raises(Exception):
            ctx.run(np.array([[0, 2]], np.uint32), 0.8)               # image out of range
        ctx.close()


def test_reference_known_answer_vectors():
    """matching/metric_test.cpp:41-129 (Metric.HAMMING_BITSET, ..._RAW_MEMORY_64BITS, ..._32BITS): the reference's own
    ground-truth Hamming distances, on the C restatement of Hamming<unsigned char>"""
    import ctypes as C
    P = _oracle.port()
    P.oracle_hamming_u8.restype = C.c_uint; P.oracle_hamming_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]

    def ham(a, b):
        a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
        return P.oracle_hamming_u8(a.ctypes.data, b.ctypes.data, len(a))

    a, b, c = (np.array([int(s, 2)], np.uint8) for s in ("01010101", "10101010", "11010100"))
    assert (ham(a, b), ham(a, a), ham(a, c)) == (8, 0, 2)
    for nbits, gt in ((64, [0, 32, 32, 33, 32, 0, 32, 21, 32, 32, 0, 31, 33, 21, 31, 0]),
                      (32, [0, 16, 16, 17, 16, 0, 16, 11, 16, 16, 0, 17, 17, 11, 17, 0])):
        i = np.arange(nbits)
        tab = [np.zeros(nbits, bool), i % 2 == 0, (i // 2) % 2 == 0, (i // 3) % 2 == 0]
        packed = [np.packbits(t, bitorder="little") for t in tab]   # std::bitset bit i = bit i of the little-endian word
        got = [ham(packed[r], packed[q]) for r in range(4) for q in range(4)]
        assert got == gt and got == [ham(packed[q], packed[r]) for r in range(4) for q in range(4)]
