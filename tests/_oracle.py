"""ctypes access to the CPU checkers under oracle/ (test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PORT_SO = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
REF_MATCH_SO = os.path.join(ROOT, "oracle", "_ref", "libref_match.so")
REF_BA_SO = os.path.join(ROOT, "oracle", "_ref", "libref_ba.so")

_port = None
_refm = None

SINK = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32)


def ensure_port():
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("match_oracle.c", "ba_oracle.cpp")]
    srcs = [s for s in srcs if os.path.exists(s)]
    if (not os.path.exists(PORT_SO)) or any(os.path.getmtime(s) > os.path.getmtime(PORT_SO) for s in srcs):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "port"], check=True, stdout=subprocess.DEVNULL)
    return PORT_SO


def port():
    global _port
    if _port is None:
        ensure_port()
        L = C.CDLL(PORT_SO)
        L.oracle_l2_u8.restype = C.c_int
        L.oracle_l2_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.oracle_l2_i32.restype = C.c_int
        L.oracle_l2_i32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.oracle_l2_f32.restype = C.c_float
        L.oracle_l2_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.oracle_search_neighbours_u8.restype = C.c_int
        L.oracle_search_neighbours_u8.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_match_distance_ratio_u8.restype = C.c_uint32
        L.oracle_match_distance_ratio_u8.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.oracle_matcher_regions_match_u8.restype = C.c_uint64
        L.oracle_matcher_regions_match_u8.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32,
                                                      C.c_void_p, C.c_uint64, C.c_float, C.c_void_p, C.c_void_p, C.c_uint64]
        L.oracle_num_threads.restype = C.c_int
        _port = L
    return _port


def have_ref_match():
    return os.path.exists(REF_MATCH_SO)


def ref_match():
    global _refm
    if _refm is None:
        L = C.CDLL(REF_MATCH_SO)
        L.ref_l2_u8.restype = C.c_int
        L.ref_l2_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.ref_uses_avx2.restype = C.c_int
        L.ref_search_neighbours_u8.restype = C.c_int
        L.ref_search_neighbours_u8.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_matcher_regions_match_u8.restype = C.c_uint64
        L.ref_matcher_regions_match_u8.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_void_p,
                                                   C.c_uint64, C.c_float, SINK, C.c_void_p]
        _refm = L
    return _refm


def _desc_tables(descs):
    arrs = [np.ascontiguousarray(d, dtype=np.uint8).reshape(-1, 128) for d in descs]
    n = len(arrs)
    ptrs = (C.c_void_p * max(n, 1))()
    cnt = (C.c_uint32 * max(n, 1))()
    for k, a in enumerate(arrs):
        ptrs[k] = a.ctypes.data if a.shape[0] else None
        cnt[k] = a.shape[0]
    return arrs, ptrs, cnt


def port_matcher_regions_match(descs, pairs, dist_ratio):
    """C-restatement oracle of Matcher_Regions::Match. Returns (offsets uint64[n_pairs+1], ij uint32[(n,2)])."""
    arrs, ptrs, cnt = _desc_tables(descs)
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
    cap = int(sum(int(arrs[j].shape[0]) for j in pairs[:, 1])) + 1 if len(pairs) else 1
    offsets = np.zeros(len(pairs) + 1, np.uint64)
    ij = np.zeros((cap, 2), np.uint32)
    total = port().oracle_matcher_regions_match_u8(ptrs, cnt, len(arrs), 128, pairs.ctypes.data, len(pairs),
                                                   np.float32(dist_ratio), offsets.ctypes.data, ij.ctypes.data, cap)
    assert total != 2 ** 64 - 1
    return offsets, ij[: int(total)].copy()


def ref_matcher_regions_match(descs, pairs, dist_ratio):
    """The reference's own Matcher_Regions(BRUTE_FORCE_L2).Match. Returns {(I, J): (n,2) uint32}."""
    arrs, ptrs, cnt = _desc_tables(descs)
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
    out = {}

    def sink(_user, I, J, pij, n):
        out[(int(I), int(J))] = np.ctypeslib.as_array(pij, shape=(int(n), 2)).copy()

    cb = SINK(sink)
    ref_match().ref_matcher_regions_match_u8(ptrs, cnt, len(arrs), pairs.ctypes.data, len(pairs), np.float32(dist_ratio), cb, None)
    return out


def offsets_to_dict(pairs, offsets, ij):
    out = {}
    for k, (a, b) in enumerate(np.asarray(pairs).reshape(-1, 2)):
        lo, hi = int(offsets[k]), int(offsets[k + 1])
        if hi > lo:
            out[(int(a), int(b))] = ij[lo:hi].copy()
    return out
