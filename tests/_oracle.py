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


def _bind_match_shim(L):
    """Prototypes of oracle/ref_shim_match.cpp (the same caller code is linked into oracle/_ref/libref_match.so and,
    against the MI355X replacements, into tests/native/_build/libmvgx_openmvg_adapter.so)."""
    L.ref_matcher_regions_match_u8.restype = C.c_uint64
    L.ref_matcher_regions_match_u8.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_void_p,
                                               C.c_uint64, C.c_float, SINK, C.c_void_p]
    return L


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


def _desc_tables(descs, dim=128):
    arrs = [np.ascontiguousarray(d, dtype=np.uint8).reshape(-1, dim) for d in descs]
    n = len(arrs)
    ptrs = (C.c_void_p * max(n, 1))()
    cnt = (C.c_uint32 * max(n, 1))()
    for k, a in enumerate(arrs):
        ptrs[k] = a.ctypes.data if a.shape[0] else None
        cnt[k] = a.shape[0]
    return arrs, ptrs, cnt


def port_matcher_regions_match(descs, pairs, dist_ratio, dim=128):
    """C-restatement oracle of Matcher_Regions::Match. Returns (offsets uint64[n_pairs+1], ij uint32[(n,2)])."""
    arrs, ptrs, cnt = _desc_tables(descs, dim)
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
    cap = int(sum(int(arrs[j].shape[0]) for j in pairs[:, 1])) + 1 if len(pairs) else 1
    offsets = np.zeros(len(pairs) + 1, np.uint64)
    ij = np.zeros((cap, 2), np.uint32)
    total = port().oracle_matcher_regions_match_u8(ptrs, cnt, len(arrs), dim, pairs.ctypes.data, len(pairs),
                                                   np.float32(dist_ratio), offsets.ctypes.data, ij.ctypes.data, cap)
    assert total != 2 ** 64 - 1
    return offsets, ij[: int(total)].copy()


def ref_matcher_regions_match(descs, pairs, dist_ratio, lib=None):
    """The reference's own Matcher_Regions(BRUTE_FORCE_L2).Match. Returns {(I, J): (n,2) uint32}.
    lib: another library exporting the same shim (the adapter build) — default oracle/_ref/libref_match.so."""
    arrs, ptrs, cnt = _desc_tables(descs)
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
    out = {}

    def sink(_user, I, J, pij, n):
        out[(int(I), int(J))] = np.ctypeslib.as_array(pij, shape=(int(n), 2)).copy()

    cb = SINK(sink)
    n = (lib or ref_match()).ref_matcher_regions_match_u8(ptrs, cnt, len(arrs), pairs.ctypes.data, len(pairs), np.float32(dist_ratio), cb, None)
    if n == 2 ** 64 - 1:
        raise RuntimeError("Matcher_Regions::Match threw")
    return out


# ---- cascade hashing ------------------------------------------------------------------------------------------------
def ref_cascade_hash(descs, lib=None):
    """The reference's hashing stage (CascadeHasher::Init(128), zero-mean descriptor over the images, CreateHashedDescriptions):
    per image (hash codes (n, 16) uint8, bucket ids (n, 6) uint16). oracle/_ref only."""
    arrs, ptrs, cnt = _desc_tables(descs)
    n = len(arrs)
    hashes = [np.zeros((len(a), 16), np.uint8) for a in arrs]
    bids = [np.zeros((len(a), 6), np.uint16) for a in arrs]
    hp = (C.c_void_p * max(n, 1))(); bp = (C.c_void_p * max(n, 1))()
    for k in range(n):
        hp[k] = hashes[k].ctypes.data if len(arrs[k]) else None
        bp[k] = bids[k].ctypes.data if len(arrs[k]) else None
    L = lib or ref_match()
    L.ref_cascade_hash_u8.restype = C.c_int
    assert L.ref_cascade_hash_u8(ptrs, cnt, n, hp, bp) == 1
    return hashes, bids


def ref_cascade_zero_mean(descs, lib=None):
    """The reference's zero-mean descriptor of the hashing stage (128 float32). oracle/_ref only."""
    arrs, ptrs, cnt = _desc_tables(descs)
    out = np.zeros(128, np.float32)
    L = lib or ref_match()
    L.ref_cascade_zero_mean_u8.restype = C.c_int
    assert L.ref_cascade_zero_mean_u8(ptrs, cnt, len(arrs), out.ctypes.data_as(C.c_void_p)) == 1
    return out


def ref_cascade_matcher_regions_match(descs, feats_xy, pairs, dist_ratio, lib=None):
    """The reference's Cascade_Hashing_Matcher_Regions(dist_ratio).Match on in-memory SIFT_Regions with the given feature
    positions. Returns {(I, J): (n, 2) uint32}. lib: the adapter build exporting the same shim."""
    arrs, ptrs, cnt = _desc_tables(descs)
    n = len(arrs)
    xy = [np.ascontiguousarray(f, np.float32).reshape(-1, 2) for f in feats_xy]
    xp = (C.c_void_p * max(n, 1))()
    for k in range(n):
        xp[k] = xy[k].ctypes.data if len(xy[k]) else None
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
    out = {}

    def sink(_user, I, J, pij, m):
        out[(int(I), int(J))] = np.ctypeslib.as_array(pij, shape=(int(m), 2)).copy()

    cb = SINK(sink)
    L = lib or ref_match()
    L.ref_cascade_matcher_regions_match_u8.restype = C.c_uint64
    L.ref_cascade_matcher_regions_match_u8(ptrs, xp, cnt, n, C.c_void_p(pairs.ctypes.data), C.c_uint64(len(pairs)), C.c_float(dist_ratio), cb, None)
    return out


def ref_cascade_matcher_regions_match_typed(kind, descs, feats_xy, pairs, dist_ratio, lib=None):
    """... the same on in-memory AKAZE_Float_Regions (kind "float64": (n, 64) float32 rows) or AKAZE_Liop_Regions (kind
    "liop144": (n, 144) uint8 rows): the other scalar region types Cascade_Hashing_Matcher_Regions::Match dispatches on."""
    dt, dim = {"float64": (np.float32, 64), "liop144": (np.uint8, 144)}[kind]
    arrs = [np.ascontiguousarray(d, dtype=dt).reshape(-1, dim) for d in descs]
    n = len(arrs)
    ptrs = (C.c_void_p * max(n, 1))()
    cnt = (C.c_uint32 * max(n, 1))()
    xy = [np.ascontiguousarray(f, np.float32).reshape(-1, 2) for f in feats_xy]
    xp = (C.c_void_p * max(n, 1))()
    for k in range(n):
        ptrs[k] = arrs[k].ctypes.data if len(arrs[k]) else None
        cnt[k] = len(arrs[k])
        xp[k] = xy[k].ctypes.data if len(xy[k]) else None
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
    out = {}

    def sink(_user, I, J, pij, m):
        out[(int(I), int(J))] = np.ctypeslib.as_array(pij, shape=(int(m), 2)).copy() if m else np.zeros((0, 2), np.uint32)

    cb = SINK(sink)
    L = lib or ref_match()
    f = getattr(L, "ref_cascade_matcher_regions_match_" + kind)
    f.restype = C.c_uint64
    f(ptrs, xp, cnt, n, C.c_void_p(pairs.ctypes.data), C.c_uint64(len(pairs)), C.c_float(dist_ratio), cb, None)
    return out


def ref_cascade_match_pair(descI, descJ, dist_ratio, n_groups=6, bits_per_bucket=10):
    """The reference's CascadeHasher with a chosen bucket layout on one pair (oracle/_ref): returns (matches (n, 2) before the
    de-duplication steps, hashI, bidsI, hashJ, bidsJ)."""
    descI = np.ascontiguousarray(descI, np.uint8).reshape(-1, 128); descJ = np.ascontiguousarray(descJ, np.uint8).reshape(-1, 128)
    hI = np.zeros((len(descI), 16), np.uint8); hJ = np.zeros((len(descJ), 16), np.uint8)
    bI = np.zeros((len(descI), n_groups), np.uint16); bJ = np.zeros((len(descJ), n_groups), np.uint16)
    out = np.zeros((max(len(descJ), 1), 2), np.uint32)
    L = ref_match()
    L.ref_cascade_match_pair_u8.restype = C.c_uint32
    n = L.ref_cascade_match_pair_u8(C.c_void_p(descI.ctypes.data), C.c_uint32(len(descI)), C.c_void_p(descJ.ctypes.data), C.c_uint32(len(descJ)),
                                    C.c_uint32(n_groups), C.c_uint32(bits_per_bucket), C.c_float(dist_ratio), C.c_void_p(hI.ctypes.data),
                                    C.c_void_p(bI.ctypes.data), C.c_void_p(hJ.ctypes.data), C.c_void_p(bJ.ctypes.data), C.c_void_p(out.ctypes.data))
    return out[: int(n)].copy(), hI, bI, hJ, bJ


def port_cascade_match_pair(descI, hashI, bidsI, descJ, hashJ, bidsJ, dist_ratio, n_groups=6, bits_per_bucket=10):
    """C restatement of the cascade MATCHING stage for one pair (queries = J, database = I), list before the reference's
    de-duplication steps: (n, 2) uint32 (descriptor of I, descriptor of J) in ascending J."""
    descI = np.ascontiguousarray(descI, np.uint8).reshape(-1, 128); descJ = np.ascontiguousarray(descJ, np.uint8).reshape(-1, 128)
    hashI = np.ascontiguousarray(hashI, np.uint8).reshape(-1, 16); hashJ = np.ascontiguousarray(hashJ, np.uint8).reshape(-1, 16)
    bidsI = np.ascontiguousarray(bidsI, np.uint16).reshape(-1, n_groups); bidsJ = np.ascontiguousarray(bidsJ, np.uint16).reshape(-1, n_groups)
    out = np.zeros((max(len(descJ), 1), 2), np.uint32)
    L = port()
    L.oracle_cascade_match_pair_u8.restype = C.c_uint32
    r = np.float32(dist_ratio)
    n = L.oracle_cascade_match_pair_u8(C.c_void_p(descI.ctypes.data), C.c_void_p(hashI.ctypes.data), C.c_void_p(bidsI.ctypes.data), C.c_uint32(len(descI)),
                                       C.c_void_p(descJ.ctypes.data), C.c_void_p(hashJ.ctypes.data), C.c_void_p(bidsJ.ctypes.data), C.c_uint32(len(descJ)),
                                       C.c_uint32(128), C.c_uint32(16), C.c_uint32(n_groups), C.c_uint32(bits_per_bucket), C.c_float(r * r), C.c_void_p(out.ctypes.data))
    return out[: int(n)].copy()


def _bin_tables(descs, L):
    arrs = [np.ascontiguousarray(d, dtype=np.uint8).reshape(-1, L) for d in descs]
    n = len(arrs)
    ptrs = (C.c_void_p * max(n, 1))()
    cnt = (C.c_uint32 * max(n, 1))()
    for k, a in enumerate(arrs):
        ptrs[k] = a.ctypes.data if a.shape[0] else None
        cnt[k] = a.shape[0]
    return arrs, ptrs, cnt


def port_matcher_regions_match_hamming(descs, pairs, dist_ratio, L=64):
    """C-restatement oracle of Matcher_Regions::Match for BRUTE_FORCE_HAMMING on L-byte binary descriptors."""
    Lb = port()
    Lb.oracle_matcher_regions_match_hamming.restype = C.c_uint64
    Lb.oracle_matcher_regions_match_hamming.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32,
                                                        C.c_void_p, C.c_uint64, C.c_float, C.c_void_p, C.c_void_p, C.c_uint64]
    arrs, ptrs, cnt = _bin_tables(descs, L)
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
    cap = int(sum(int(arrs[j].shape[0]) for j in pairs[:, 1])) + 1 if len(pairs) else 1
    offsets = np.zeros(len(pairs) + 1, np.uint64)
    ij = np.zeros((cap, 2), np.uint32)
    total = Lb.oracle_matcher_regions_match_hamming(ptrs, cnt, len(arrs), L, pairs.ctypes.data, len(pairs),
                                                    np.float32(dist_ratio), offsets.ctypes.data, ij.ctypes.data, cap)
    assert total != 2 ** 64 - 1
    return offsets, ij[: int(total)].copy()


def ref_matcher_regions_match_binary64(descs, pairs, dist_ratio, lib=None):
    """The reference's own Matcher_Regions(BRUTE_FORCE_HAMMING).Match on AKAZE_Binary_Regions. -> {(I, J): (n,2) uint32}"""
    Lr = lib or ref_match()
    Lr.ref_matcher_regions_match_binary64.restype = C.c_uint64
    Lr.ref_matcher_regions_match_binary64.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_void_p,
                                                      C.c_uint64, C.c_float, SINK, C.c_void_p]
    arrs, ptrs, cnt = _bin_tables(descs, 64)
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
    out = {}

    def sink(_user, I, J, pij, n):
        out[(int(I), int(J))] = np.ctypeslib.as_array(pij, shape=(int(n), 2)).copy()

    cb = SINK(sink)
    Lr.ref_matcher_regions_match_binary64(ptrs, cnt, len(arrs), pairs.ctypes.data, len(pairs), np.float32(dist_ratio), cb, None)
    return out


def _f32_tables(descs, dim):
    arrs = [np.ascontiguousarray(d, dtype=np.float32).reshape(-1, dim) for d in descs]
    n = len(arrs)
    ptrs = (C.c_void_p * max(n, 1))()
    cnt = (C.c_uint32 * max(n, 1))()
    for k, a in enumerate(arrs):
        ptrs[k] = a.ctypes.data if a.shape[0] else None
        cnt[k] = a.shape[0]
    return arrs, ptrs, cnt


def port_matcher_regions_match_f32(descs, pairs, dist_ratio, dim=64):
    """C-restatement oracle of Matcher_Regions::Match for BRUTE_FORCE_L2 on float descriptors."""
    Lb = port()
    Lb.oracle_matcher_regions_match_f32.restype = C.c_uint64
    Lb.oracle_matcher_regions_match_f32.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32,
                                                    C.c_void_p, C.c_uint64, C.c_float, C.c_void_p, C.c_void_p, C.c_uint64]
    arrs, ptrs, cnt = _f32_tables(descs, dim)
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
    cap = int(sum(int(arrs[j].shape[0]) for j in pairs[:, 1])) + 1 if len(pairs) else 1
    offsets = np.zeros(len(pairs) + 1, np.uint64)
    ij = np.zeros((cap, 2), np.uint32)
    total = Lb.oracle_matcher_regions_match_f32(ptrs, cnt, len(arrs), dim, pairs.ctypes.data, len(pairs),
                                                np.float32(dist_ratio), offsets.ctypes.data, ij.ctypes.data, cap)
    assert total != 2 ** 64 - 1
    return offsets, ij[: int(total)].copy()


def ref_matcher_regions_match_float64(descs, pairs, dist_ratio, lib=None):
    """The reference's own Matcher_Regions(BRUTE_FORCE_L2).Match on AKAZE_Float_Regions. -> {(I, J): (n,2) uint32}"""
    Lr = lib or ref_match()
    Lr.ref_matcher_regions_match_float64.restype = C.c_uint64
    Lr.ref_matcher_regions_match_float64.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_void_p,
                                                     C.c_uint64, C.c_float, SINK, C.c_void_p]
    arrs, ptrs, cnt = _f32_tables(descs, 64)
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
    out = {}

    def sink(_user, I, J, pij, n):
        out[(int(I), int(J))] = np.ctypeslib.as_array(pij, shape=(int(n), 2)).copy()

    cb = SINK(sink)
    Lr.ref_matcher_regions_match_float64(ptrs, cnt, len(arrs), pairs.ctypes.data, len(pairs), np.float32(dist_ratio), cb, None)
    return out


def ref_matcher_regions_match_liop144(descs, pairs, dist_ratio, lib=None):
    """The reference's own Matcher_Regions(BRUTE_FORCE_L2).Match on AKAZE_Liop_Regions (144 x uint8). -> {(I, J): (n,2)}"""
    Lr = lib or ref_match()
    Lr.ref_matcher_regions_match_liop144.restype = C.c_uint64
    Lr.ref_matcher_regions_match_liop144.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_void_p,
                                                     C.c_uint64, C.c_float, SINK, C.c_void_p]
    arrs, ptrs, cnt = _desc_tables(descs, 144)
    pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
    out = {}

    def sink(_user, I, J, pij, n):
        out[(int(I), int(J))] = np.ctypeslib.as_array(pij, shape=(int(n), 2)).copy()

    cb = SINK(sink)
    Lr.ref_matcher_regions_match_liop144(ptrs, cnt, len(arrs), pairs.ctypes.data, len(pairs), np.float32(dist_ratio), cb, None)
    return out


def offsets_to_dict(pairs, offsets, ij):
    out = {}
    for k, (a, b) in enumerate(np.asarray(pairs).reshape(-1, 2)):
        lo, hi = int(offsets[k]), int(offsets[k + 1])
        if hi > lo:
            out[(int(a), int(b))] = ij[lo:hi].copy()
    return out


# ---------------------------------------------------------------------------------------------------------
# bundle adjustment checkers
# ---------------------------------------------------------------------------------------------------------
_refba = None


def have_ref_ba():
    return os.path.exists(REF_BA_SO)


def ba_problem_struct(scene, pose_const_mask=None, intr_const_mask=None, points_constant=False, huber_a=None):
    """Builds a ctypes mvgx_ba_problem (+ the dict of arrays that must stay alive)."""
    from openmvg_amd import _capi
    keep = {}

    def arr(name, dtype):
        a = np.ascontiguousarray(scene[name], dtype=dtype)
        keep[name] = a
        return a.ctypes.data

    p = _capi.BaProblem()
    p.n_poses = int(scene["n_poses"]); p.n_intrinsics = int(scene["n_intrinsics"]); p.n_points = int(scene["n_points"])
    p.n_obs = int(scene["n_obs"])
    p.poses = arr("poses", np.float64); p.intrinsics = arr("intrinsics", np.float64)
    p.intr_model = arr("intr_model", np.int32); p.points = arr("points", np.float64)
    p.obs_pose = arr("obs_pose", np.uint32); p.obs_intr = arr("obs_intr", np.uint32); p.obs_point = arr("obs_point", np.uint32)
    p.obs_xy = arr("obs_xy", np.float64)
    if pose_const_mask is not None:
        keep["pm"] = np.ascontiguousarray(pose_const_mask, np.uint8); p.pose_const_mask = keep["pm"].ctypes.data
    if intr_const_mask is not None:
        keep["im"] = np.ascontiguousarray(intr_const_mask, np.uint8); p.intr_const_mask = keep["im"].ctypes.data
    p.points_constant = 1 if points_constant else 0
    p.huber_a = float(scene.get("huber_a", 16.0) if huber_a is None else huber_a)
    if scene.get("obs_weight") is not None:
        p.obs_weight = arr("obs_weight", np.float64)
    if scene.get("obs_is_control") is not None:
        p.obs_is_control = arr("obs_is_control", np.uint8)
    if scene.get("point_const_mask") is not None:
        p.point_const_mask = arr("point_const_mask", np.uint8)
    if scene.get("prior_pose") is not None and len(scene["prior_pose"]):
        p.n_pose_priors = len(scene["prior_pose"])
        p.prior_pose = arr("prior_pose", np.uint32)
        p.prior_center = arr("prior_center", np.float64)
        p.prior_weight = arr("prior_weight", np.float64)
        p.prior_huber_a = float(scene.get("prior_huber_a", 0.0))
    return p, keep


def default_ba_options(**kw):
    from openmvg_amd import _capi
    o = _capi.BaOptions(50, 1e-6, 1e-10, 1e-8, 1e4, 1e16, 1e-32, 1e-3, 1e-6, 1e32, 5, 1, 0)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def port_ba_solve(scene, options=None, trace_cap=64, **pk):
    from openmvg_amd import _capi
    L = port()
    L.oracle_ba_solve.restype = C.c_int
    L.oracle_ba_solve.argtypes = [C.POINTER(_capi.BaProblem), C.POINTER(_capi.BaOptions), C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.POINTER(_capi.BaSummary), C.c_void_p, C.c_int]
    prob, keep = ba_problem_struct(scene, **pk)
    opt = options or default_ba_options()
    poses = np.zeros_like(keep["poses"]); intr = np.zeros_like(keep["intrinsics"]); pts = np.zeros_like(keep["points"])
    summ = _capi.BaSummary()
    trace = np.zeros((trace_cap, 6))
    rc = L.oracle_ba_solve(C.byref(prob), C.byref(opt), poses.ctypes.data, intr.ctypes.data, pts.ctypes.data, C.byref(summ),
                           trace.ctypes.data, trace_cap)
    return rc, summ, poses, intr, pts, trace[: max(0, summ.num_iterations)]


def port_ba_evaluate(scene, **pk):
    from openmvg_amd import _capi
    L = port()
    L.oracle_ba_evaluate.restype = C.c_int
    L.oracle_ba_evaluate.argtypes = [C.POINTER(_capi.BaProblem), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    prob, keep = ba_problem_struct(scene, **pk)
    cost, rmse = C.c_double(), C.c_double()
    rc = L.oracle_ba_evaluate(C.byref(prob), C.byref(cost), C.byref(rmse))
    assert rc == 0
    return cost.value, rmse.value


def port_ba_eval_obs(model, intr, pose, X, obs):
    L = port()
    L.oracle_ba_eval_obs.restype = C.c_int
    L.oracle_ba_eval_obs.argtypes = [C.c_int] + [C.c_void_p] * 8
    a = [np.ascontiguousarray(v, np.float64) for v in (intr, pose, X, obs)]
    r = np.zeros(2); Ji = np.zeros((2, 8)); Jc = np.zeros((2, 6)); Jp = np.zeros((2, 3))
    rc = L.oracle_ba_eval_obs(int(model), a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data,
                              r.ctypes.data, Ji.ctypes.data, Jc.ctypes.data, Jp.ctypes.data)
    assert rc == 0
    return r, Ji, Jc, Jp


class ShimExtras(C.Structure):
    """struct Extras of oracle/ref_shim_ba.cpp"""
    _fields_ = [("n_ctrl_points", C.c_uint32), ("ctrl_X", C.c_void_p), ("n_ctrl_obs", C.c_uint64), ("ctrl_obs_pose", C.c_void_p),
                ("ctrl_obs_point", C.c_void_p), ("ctrl_obs_xy", C.c_void_p), ("ctrl_weight", C.c_double),
                ("use_control_points", C.c_int), ("prior_flag", C.c_void_p), ("prior_center", C.c_void_p),
                ("prior_weight", C.c_void_p), ("use_motion_priors", C.c_int)]


def port_ba_eval_prior(pose, center, weight):
    L = port()
    L.oracle_ba_eval_prior.restype = C.c_int
    L.oracle_ba_eval_prior.argtypes = [C.c_void_p] * 5
    a = [np.ascontiguousarray(v, np.float64) for v in (pose, center, weight)]
    r = np.zeros(3); Jc = np.zeros((3, 6))
    assert L.oracle_ba_eval_prior(a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, r.ctypes.data, Jc.ctypes.data) == 0
    return r, Jc


def _bind_ba_shim(L):
    L.ref_ba_adjust_ex.restype = C.c_int
    L.ref_ba_adjust_ex.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(ShimExtras), C.c_void_p]
    L.ref_ba_prior_prepare.restype = C.c_int
    L.ref_ba_prior_prepare.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64] + [C.c_void_p] * 12
    L.ref_ba_adjust.restype = C.c_int
    L.ref_ba_adjust.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    return L


def ref_ba_adjust(scene, intrinsics_opt=None, extrinsics_opt=6, structure_opt=1, max_iterations=0, num_threads=0,
                  linear_solver=0, use_loss=1, print_summary=0, lib=None):
    """The reference's Bundle_Adjustment_Ceres::Adjust. Returns (rc, stats[4], poses, intrinsics, points).
    intrinsics_opt default = ADJUST_ALL (cameras/Camera_Common.hpp:92-100: focal 2 | pp 4 | disto 8 = 14)."""
    global _refba
    if lib is None and _refba is None:
        _refba = _bind_ba_shim(C.CDLL(REF_BA_SO))
    poses = np.ascontiguousarray(scene["poses"], np.float64).copy()
    intr = np.ascontiguousarray(scene["intrinsics"], np.float64).copy()
    pts = np.ascontiguousarray(scene["points"], np.float64).copy()
    model = np.ascontiguousarray(scene["intr_model"], np.int32)
    op = np.ascontiguousarray(scene["obs_pose"], np.uint32); oi = np.ascontiguousarray(scene["obs_intr"], np.uint32)
    ox = np.ascontiguousarray(scene["obs_point"], np.uint32); xy = np.ascontiguousarray(scene["obs_xy"], np.float64)
    stats = np.zeros(4)
    if intrinsics_opt is None:
        intrinsics_opt = 14
    rc = (lib or _refba).ref_ba_adjust(int(scene["n_poses"]), int(scene["n_intrinsics"]), int(scene["n_points"]), int(scene["n_obs"]),
                              poses.ctypes.data, intr.ctypes.data, model.ctypes.data, pts.ctypes.data, op.ctypes.data,
                              oi.ctypes.data, ox.ctypes.data, xy.ctypes.data, int(intrinsics_opt), int(extrinsics_opt),
                              int(structure_opt), int(max_iterations), int(num_threads), int(linear_solver), int(use_loss),
                              int(print_summary), stats.ctypes.data)
    return rc, stats, poses, intr, pts


def _structure_part(scene):
    """Splits a flat problem into the Landmark part and the control-point part (appended by synth.add_control_points)."""
    ns = int(scene.get("n_structure_points", scene["n_points"]))
    ctrl = np.asarray(scene["obs_is_control"], bool) if scene.get("obs_is_control") is not None else np.zeros(int(scene["n_obs"]), bool)
    return ns, ctrl


def ref_ba_adjust_ex(scene, intrinsics_opt=14, extrinsics_opt=6, structure_opt=1, max_iterations=0, num_threads=0, linear_solver=0,
                     use_loss=1, use_control_points=None, use_motion_priors=None, lib=None):
    """Bundle_Adjustment_Ceres::Adjust on a flat problem that may carry control points (synth.add_control_points) and pose-centre
    priors (synth.add_pose_priors; the reference derives the priors' Huber scale itself). Returns (rc, stats, poses, intr, points)
    with `points` in the flat layout (control points unchanged at the end)."""
    global _refba
    if lib is None and _refba is None:
        _refba = _bind_ba_shim(C.CDLL(REF_BA_SO))
    L = lib or _refba
    ns, ctrl = _structure_part(scene)
    poses = np.ascontiguousarray(scene["poses"], np.float64).copy()
    intr = np.ascontiguousarray(scene["intrinsics"], np.float64).copy()
    pts_all = np.ascontiguousarray(scene["points"], np.float64).copy()
    pts = np.ascontiguousarray(pts_all[:ns])
    model = np.ascontiguousarray(scene["intr_model"], np.int32)
    op = np.ascontiguousarray(np.asarray(scene["obs_pose"])[~ctrl], np.uint32); oi = np.ascontiguousarray(np.asarray(scene["obs_intr"])[~ctrl], np.uint32)
    ox = np.ascontiguousarray(np.asarray(scene["obs_point"])[~ctrl], np.uint32); xy = np.ascontiguousarray(np.asarray(scene["obs_xy"])[~ctrl], np.float64)
    ex = ShimExtras()
    keep = []
    if ctrl.any():
        cX = np.ascontiguousarray(pts_all[ns:]); cop = np.ascontiguousarray(np.asarray(scene["obs_pose"])[ctrl], np.uint32)
        cox = np.ascontiguousarray(np.asarray(scene["obs_point"])[ctrl] - ns, np.uint32); cxy = np.ascontiguousarray(np.asarray(scene["obs_xy"])[ctrl], np.float64)
        keep += [cX, cop, cox, cxy]
        ex.n_ctrl_points = len(cX); ex.ctrl_X = cX.ctypes.data; ex.n_ctrl_obs = len(cop); ex.ctrl_obs_pose = cop.ctypes.data
        ex.ctrl_obs_point = cox.ctypes.data; ex.ctrl_obs_xy = cxy.ctypes.data; ex.ctrl_weight = float(scene.get("control_weight", 20.0))
        ex.use_control_points = 1 if use_control_points is None else int(use_control_points)
    if scene.get("prior_pose") is not None and len(scene["prior_pose"]):
        n = int(scene["n_poses"])
        flag = np.zeros(n, np.uint8); pc = np.zeros((n, 3)); pw = np.zeros((n, 3))
        flag[scene["prior_pose"]] = 1; pc[scene["prior_pose"]] = scene["prior_center"]; pw[scene["prior_pose"]] = scene["prior_weight"]
        keep += [flag, pc, pw]
        ex.prior_flag = flag.ctypes.data; ex.prior_center = pc.ctypes.data; ex.prior_weight = pw.ctypes.data
        ex.use_motion_priors = 1 if use_motion_priors is None else int(use_motion_priors)
    stats = np.zeros(4)
    rc = L.ref_ba_adjust_ex(int(scene["n_poses"]), int(scene["n_intrinsics"]), ns, len(op), poses.ctypes.data, intr.ctypes.data,
                            model.ctypes.data, pts.ctypes.data, op.ctypes.data, oi.ctypes.data, ox.ctypes.data, xy.ctypes.data,
                            int(intrinsics_opt), int(extrinsics_opt), int(structure_opt), int(max_iterations), int(num_threads),
                            int(linear_solver), int(use_loss), 0, C.byref(ex), stats.ctypes.data)
    pts_all[:ns] = pts
    return rc, stats, poses, intr, pts_all


def ref_ba_prior_prepare(scene, lib=None):
    """The scene transformation the reference applies before building the problem when motion priors are on
    (oracle/ref_shim_ba.cpp::ref_ba_prior_prepare). Returns (usable, new_scene, centroid): new_scene carries the transformed
    poses / points / prior centres and prior_huber_a = Square(pose_center_robust_fitting_error)."""
    global _refba
    if lib is None and _refba is None:
        _refba = _bind_ba_shim(C.CDLL(REF_BA_SO))
    L = lib or _refba
    n = int(scene["n_poses"])
    poses = np.ascontiguousarray(scene["poses"], np.float64).copy(); intr = np.ascontiguousarray(scene["intrinsics"], np.float64).copy()
    pts = np.ascontiguousarray(scene["points"], np.float64).copy(); model = np.ascontiguousarray(scene["intr_model"], np.int32)
    op = np.ascontiguousarray(scene["obs_pose"], np.uint32); oi = np.ascontiguousarray(scene["obs_intr"], np.uint32)
    ox = np.ascontiguousarray(scene["obs_point"], np.uint32); xy = np.ascontiguousarray(scene["obs_xy"], np.float64)
    flag = np.zeros(n, np.uint8); pc = np.zeros((n, 3)); pw = np.zeros((n, 3))
    flag[scene["prior_pose"]] = 1; pc[scene["prior_pose"]] = scene["prior_center"]; pw[scene["prior_pose"]] = scene["prior_weight"]
    out = np.zeros(5)
    rc = L.ref_ba_prior_prepare(n, int(scene["n_intrinsics"]), int(scene["n_points"]), int(scene["n_obs"]), poses.ctypes.data,
                                intr.ctypes.data, model.ctypes.data, pts.ctypes.data, op.ctypes.data, oi.ctypes.data, ox.ctypes.data,
                                xy.ctypes.data, flag.ctypes.data, pc.ctypes.data, pw.ctypes.data, out.ctypes.data)
    assert rc == 0
    sc = dict(scene)
    sc["poses"] = poses; sc["points"] = pts
    sc["prior_center"] = np.ascontiguousarray(pc[scene["prior_pose"]])
    sc["prior_huber_a"] = float(out[1]) ** 2
    return bool(out[0]), sc, out[2:5].copy()


def _flat(scene):
    return (np.ascontiguousarray(scene["poses"], np.float64), np.ascontiguousarray(scene["intrinsics"], np.float64),
            np.ascontiguousarray(scene["intr_model"], np.int32), np.ascontiguousarray(scene["points"], np.float64),
            np.ascontiguousarray(scene["obs_pose"], np.uint32), np.ascontiguousarray(scene["obs_intr"], np.uint32),
            np.ascontiguousarray(scene["obs_point"], np.uint32), np.ascontiguousarray(scene["obs_xy"], np.float64))


def port_ba_track_angles(scene):
    """oracle/ba_oracle.cpp::oracle_ba_track_angles: per-track maximum ray angle (degrees)."""
    L = port()
    L.oracle_ba_track_angles.restype = C.c_int
    L.oracle_ba_track_angles.argtypes = [C.c_uint32, C.c_uint64] + [C.c_void_p] * 8
    poses, intr, model, pts, op, oi, ox, xy = _flat(scene)
    out = np.zeros(len(pts))
    rc = L.oracle_ba_track_angles(len(pts), len(op), poses.ctypes.data, intr.ctypes.data, model.ctypes.data, op.ctypes.data,
                                  oi.ctypes.data, ox.ctypes.data, xy.ctypes.data, out.ctypes.data)
    assert rc == 0
    return out


def ref_ba_filters(scene, px_threshold=4.0, min_track_length=2, min_angle_deg=2.0, lib_path=None, lib=None):
    """The reference's RemoveOutliers_PixelResidualError + RemoveOutliers_AngleError (sfm/sfm_data_filters.cpp) on the flat
    scene (oracle/ref_shim_ba.cpp::ref_ba_filters). -> (obs_keep bool[n_obs], (n_residual, n_angle), max_angle[n_points]).
    lib: a loaded adapter library (adapter() / adapter_ba_emu()): the same caller code over the replacement TU
    openmvg_amd/adapter/mvgx_outlier_filters.cpp."""
    L = lib if lib is not None else C.CDLL(lib_path or REF_BA_SO)
    L.ref_ba_filters.restype = C.c_int
    L.ref_ba_filters.argtypes = ([C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64] + [C.c_void_p] * 8 +
                                 [C.c_double, C.c_uint32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p])
    poses, intr, model, pts, op, oi, ox, xy = _flat(scene)
    keep = np.zeros(len(op), np.uint8); counts = np.zeros(2, np.uint64); ang = np.zeros(len(pts))
    rc = L.ref_ba_filters(len(poses), len(intr), len(pts), len(op), poses.ctypes.data, intr.ctypes.data, model.ctypes.data,
                          pts.ctypes.data, op.ctypes.data, oi.ctypes.data, ox.ctypes.data, xy.ctypes.data, float(px_threshold),
                          int(min_track_length), float(min_angle_deg), keep.ctypes.data, counts.ctypes.data, ang.ctypes.data)
    assert rc == 0, rc
    return keep.astype(bool), (int(counts[0]), int(counts[1])), ang


def ref_ba_filters_timed(scene, px_threshold=4.0, min_track_length=2, min_angle_deg=2.0, lib=None):
    """ref_ba_filters + the wall time inside the two filter calls -> (obs_keep, (n_residual, n_angle), (s_residual, s_angle))"""
    L = lib if lib is not None else C.CDLL(REF_BA_SO)
    fn = L.ref_ba_filters_timed
    fn.restype = C.c_int
    fn.argtypes = ([C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64] + [C.c_void_p] * 8 + [C.c_double, C.c_uint32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p])
    poses, intr, model, pts, op, oi, ox, xy = _flat(scene)
    keep = np.zeros(len(op), np.uint8); counts = np.zeros(2, np.uint64); sec = np.zeros(2)
    rc = fn(len(poses), len(intr), len(pts), len(op), poses.ctypes.data, intr.ctypes.data, model.ctypes.data, pts.ctypes.data, op.ctypes.data,
            oi.ctypes.data, ox.ctypes.data, xy.ctypes.data, float(px_threshold), int(min_track_length), float(min_angle_deg), keep.ctypes.data,
            counts.ctypes.data, sec.ctypes.data)
    assert rc == 0, rc
    return keep.astype(bool), (int(counts[0]), int(counts[1])), (float(sec[0]), float(sec[1]))


def ref_ba_reject_loop(scene, px_threshold=4.0, count=0, max_rounds=8, num_threads=0, lib=None):
    """oracle/ref_shim_ba.cpp::ref_ba_reject_loop: `do { Adjust } while (badTrackRejector)` on one SfM_Data.
    -> dict(keep, poses, intrinsics, points, rounds, seconds[rounds][3], removed[rounds][2], rmse)"""
    L = lib if lib is not None else C.CDLL(REF_BA_SO)
    fn = L.ref_ba_reject_loop
    fn.restype = C.c_int
    fn.argtypes = ([C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64] + [C.c_void_p] * 8 + [C.c_double, C.c_uint32, C.c_int, C.c_int] + [C.c_void_p] * 5)
    poses, intr, model, pts, op, oi, ox, xy = _flat(scene)
    poses = poses.copy(); intr = intr.copy(); pts = pts.copy()
    keep = np.zeros(len(op), np.uint8); rounds = np.zeros(1, np.int32); sec = np.zeros(3 * max_rounds); rem = np.zeros(2 * max_rounds, np.uint64)
    rmse = np.zeros(1)
    rc = fn(len(poses), len(intr), len(pts), len(op), poses.ctypes.data, intr.ctypes.data, model.ctypes.data, pts.ctypes.data, op.ctypes.data,
            oi.ctypes.data, ox.ctypes.data, xy.ctypes.data, float(px_threshold), int(count), int(max_rounds), int(num_threads), keep.ctypes.data,
            rounds.ctypes.data, sec.ctypes.data, rem.ctypes.data, rmse.ctypes.data)
    assert rc == 0, rc
    r = int(rounds[0])
    return dict(keep=keep.astype(bool), poses=poses, intrinsics=intr, points=pts, rounds=r, seconds=sec[:3 * r].reshape(r, 3),
                removed=rem[:2 * r].reshape(r, 2).astype(np.int64), rmse=float(rmse[0]))


def ref_ba_adjust_growing(scene, n_new_tracks=0, max_iterations=0, num_threads=0, lib=None):
    """oracle/ref_shim_ba.cpp::ref_ba_adjust_growing: Adjust() without the last pose's view, then the view + its observations + n_new_tracks
    of its tracks are added to the same SfM_Data and Adjust() runs again (resection, then BA: sequential_SfM.cpp:206-210).
    -> dict(rc, seconds[2], rmse[3], counts (obs, tracks, obs, tracks), poses, intrinsics, points)"""
    L = lib if lib is not None else C.CDLL(REF_BA_SO)
    fn = L.ref_ba_adjust_growing
    fn.restype = C.c_int
    fn.argtypes = ([C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64] + [C.c_void_p] * 8 + [C.c_uint32, C.c_int, C.c_int] + [C.c_void_p] * 3)
    poses, intr, model, pts, op, oi, ox, xy = _flat(scene)
    poses = poses.copy(); intr = intr.copy(); pts = pts.copy()
    sec = np.zeros(2); rmse = np.zeros(3); counts = np.zeros(4, np.uint64)
    rc = fn(len(poses), len(intr), len(pts), len(op), poses.ctypes.data, intr.ctypes.data, model.ctypes.data, pts.ctypes.data, op.ctypes.data,
            oi.ctypes.data, ox.ctypes.data, xy.ctypes.data, int(n_new_tracks), int(max_iterations), int(num_threads), sec.ctypes.data, rmse.ctypes.data,
            counts.ctypes.data)
    return dict(rc=rc, seconds=sec, rmse=rmse, counts=counts.astype(np.int64), poses=poses, intrinsics=intr, points=pts)


def ref_save_baf(scene, path):
    """The reference's Save_BAF (sfm/sfm_data_io_baf.hpp) on the flat scene (oracle/ref_shim_ba.cpp::ref_save_baf)."""
    L = C.CDLL(REF_BA_SO)
    L.ref_save_baf.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64] + [C.c_void_p] * 8 + [C.c_char_p]
    poses = np.ascontiguousarray(scene["poses"], np.float64); intr = np.ascontiguousarray(scene["intrinsics"], np.float64)
    pts = np.ascontiguousarray(scene["points"], np.float64); model = np.ascontiguousarray(scene["intr_model"], np.int32)
    op = np.ascontiguousarray(scene["obs_pose"], np.uint32); oi = np.ascontiguousarray(scene["obs_intr"], np.uint32)
    ox = np.ascontiguousarray(scene["obs_point"], np.uint32); xy = np.ascontiguousarray(scene["obs_xy"], np.float64)
    return L.ref_save_baf(len(poses), len(intr), len(pts), len(op), poses.ctypes.data, intr.ctypes.data, model.ctypes.data,
                          pts.ctypes.data, op.ctypes.data, oi.ctypes.data, ox.ctypes.data, xy.ctypes.data, path.encode())


# ---------------------------------------------------------------------------------------------------------
# the openMVG-side adapter build (product code + the same caller shims as oracle/_ref)
# ---------------------------------------------------------------------------------------------------------
ADAPTER_SO = os.path.join(ROOT, "tests", "native", "_build", "libmvgx_openmvg_adapter.so")
ADAPTER_BA_SO = os.path.join(ROOT, "tests", "native", "_build", "libmvgx_openmvg_adapter_ba.so")
_adapter = None


def build_adapter_harness(verbose=False):
    """tests/native/adapter_harness.mk: the product's adapter objects + the reference's caller shims (needs the openMVG tree
    and the objects of oracle/Makefile; a no-op elsewhere - the GPU box uses the prebuilt libraries)."""
    import subprocess
    if not os.path.isdir("/root/reference/src"):
        return None
    subprocess.run(["make", "-f", os.path.join(ROOT, "tests", "native", "adapter_harness.mk")], check=True,
                   stdout=None if verbose else subprocess.DEVNULL)
    return ADAPTER_SO


ADAPTER_EMU_SO = os.path.join(ROOT, "tests", "native", "_build", "libmvgx_openmvg_adapter_emu.so")
_adapter_emu = None


def adapter_emu():
    """The matcher adapter linked against the HIP emulation libraries (tests/native/adapter_harness.mk, target `emu`): the
    adapter's C++ code on the CPU. None where the openMVG tree / reference objects are absent."""
    global _adapter_emu
    if _adapter_emu is None:
        import subprocess
        from tests import _emu
        if not os.path.isdir("/root/reference/src") or not os.path.exists(os.path.join(ROOT, "openmvg_amd", "lib", "adapter_obj", "mvgx_matcher_regions.o")):
            return None
        try:   # a missing prerequisite (reference objects, compiler) skips the emulated adapter tests instead of breaking collection
            _emu.build(); _emu.build_match()
            subprocess.run(["make", "-f", os.path.join(ROOT, "tests", "native", "adapter_harness.mk"), "emu"], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            _adapter_emu = _bind_match_shim(C.CDLL(ADAPTER_EMU_SO, mode=os.RTLD_LOCAL | os.RTLD_NOW | os.RTLD_DEEPBIND))
        except Exception:
            return None
    return _adapter_emu


ADAPTER_BA_EMU_SO = os.path.join(ROOT, "tests", "native", "_build", "libmvgx_openmvg_adapter_ba_emu.so")
_adapter_ba_emu = None


def adapter_ba_emu():
    """The BA adapter (Bundle_Adjustment_Ceres replacement TU + Bundle_Adjustment_HIP) linked against the HIP emulation."""
    global _adapter_ba_emu
    if _adapter_ba_emu is None:
        if adapter_emu() is None:   # builds the `emu` targets
            return None
        try:
            _adapter_ba_emu = _bind_ba_shim(C.CDLL(ADAPTER_BA_EMU_SO, mode=os.RTLD_LOCAL | os.RTLD_NOW | os.RTLD_DEEPBIND))
        except Exception:
            return None
    return _adapter_ba_emu


def have_adapter():
    return os.path.exists(ADAPTER_SO) and os.path.exists(ADAPTER_BA_SO)


def adapter():
    """libmvgx_openmvg_adapter.so: oracle/ref_shim_{match,ba}.cpp linked against openmvg_amd/adapter/*.cpp (the link-time
    replacements of Matcher_Regions.cpp / sfm_data_BA_ceres.cpp) and libmvgx_hip.so. RTLD_LOCAL + -Bsymbolic keep its
    Matcher_Regions / Bundle_Adjustment_Ceres symbols apart from the reference's in oracle/_ref."""
    global _adapter
    if _adapter is None:
        class _Both:   # the two adapter libraries behind one handle (matcher half / BA half: different Eigen ABIs)
            pass
        both = _Both()
        m = _bind_match_shim(C.CDLL(ADAPTER_SO, mode=os.RTLD_LOCAL | os.RTLD_NOW))
        b = _bind_ba_shim(C.CDLL(ADAPTER_BA_SO, mode=os.RTLD_LOCAL | os.RTLD_NOW))
        for name in ("ref_matcher_regions_match_u8", "ref_matcher_regions_match_binary64", "ref_matcher_regions_match_float64",
                     "ref_matcher_regions_match_liop144", "ref_cascade_matcher_regions_match_u8", "ref_cascade_matcher_regions_match_float64",
                     "ref_cascade_matcher_regions_match_liop144", "ref_cascade_hash_u8", "mvgx_adapter_counters",
                     "mvgx_adapter_cascade_last_hash_check"):   # (the counters of the matcher half: which route produced a container)
            setattr(both, name, getattr(m, name))
        for name in ("ref_ba_adjust", "ref_ba_adjust_ex", "ref_ba_prior_prepare", "ref_ba_filters", "ref_ba_filters_timed", "ref_ba_reject_loop", "ref_ba_adjust_growing", "mvgx_adapter_ba_context_stats", "mvgx_adapter_ba_context_stats3",
                     "mvgx_adapter_ba_release_context", "mvgx_adapter_ba_kept_solver_info"):
            setattr(both, name, getattr(b, name))
        both.ba_counters = b.mvgx_adapter_counters   # (the counters of the BA half)
        _adapter = both
    return _adapter


# ---------------------------------------------------------------------------------------------------------
# geometric filter checkers
# ---------------------------------------------------------------------------------------------------------
_refgeo = None


def have_ref_geofilter():
    return os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_geofilter.so"))


def _geofilter_call(fn, tv, precision, max_iterations, threads=None):
    xI = np.ascontiguousarray(tv["xI"], np.float64); xJ = np.ascontiguousarray(tv["xJ"], np.float64)
    start = np.ascontiguousarray(tv["start"], np.uint64); wh = np.ascontiguousarray(tv["wh"], np.uint32)
    n_pairs = len(start) - 1
    mask = np.zeros(max(int(start[-1]), 1), np.uint8); ok = np.zeros(max(n_pairs, 1), np.uint8)
    F = np.zeros((max(n_pairs, 1), 9)); prec = np.zeros(max(n_pairs, 1)); nfa = np.zeros(max(n_pairs, 1))
    P = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    args = [P(xI), P(xJ), P(start), P(wh), C.c_uint64(n_pairs), C.c_double(precision), C.c_uint32(max_iterations)]
    if threads is not None:
        args.append(C.c_int(threads))
    fn.restype = C.c_double
    secs = fn(*args, P(mask), P(ok), P(F), P(prec), P(nfa))
    return dict(mask=mask[:int(start[-1])].astype(bool), ok=ok[:n_pairs].astype(bool), F=F[:n_pairs].reshape(-1, 3, 3), precision=prec[:n_pairs],
                nfa=nfa[:n_pairs], seconds=secs)


def ref_geofilter(tv, precision=4.0, max_iterations=2048, threads=0):
    """The reference's own ACKernelAdaptor<SevenPointSolver, EpipolarDistanceError> + ACRANSAC per pair (oracle/_ref/libref_geofilter.so)."""
    global _refgeo
    if _refgeo is None:
        _refgeo = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_geofilter.so"))
    return _geofilter_call(_refgeo.ref_geofilter_f_acransac, tv, precision, max_iterations, threads)


def ref_geofilter_h(tv, precision=4.0, max_iterations=2048, threads=0):
    """The reference's ACKernelAdaptor<FourPointSolver, AsymmetricError, UnnormalizerI> (point-to-point) + ACRANSAC per pair; "F" = m_H."""
    global _refgeo
    if _refgeo is None:
        _refgeo = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_geofilter.so"))
    return _geofilter_call(_refgeo.ref_geofilter_h_acransac, tv, precision, max_iterations, threads)


def _geofilter_call_e(fn, tv, K, bearings, precision, max_iterations, threads=None):
    """the essential model: fn(xI, xJ, start, wh, K, [bI, bJ,] n_pairs, precision, max_iterations, [threads,] mask, ok, E, precision, nfa)"""
    xI = np.ascontiguousarray(tv["xI"], np.float64); xJ = np.ascontiguousarray(tv["xJ"], np.float64)
    start = np.ascontiguousarray(tv["start"], np.uint64); wh = np.ascontiguousarray(tv["wh"], np.uint32)
    K = np.ascontiguousarray(K, np.float64).reshape(-1, 18)
    n_pairs = len(start) - 1
    mask = np.zeros(max(int(start[-1]), 1), np.uint8); ok = np.zeros(max(n_pairs, 1), np.uint8)
    F = np.zeros((max(n_pairs, 1), 9)); prec = np.zeros(max(n_pairs, 1)); nfa = np.zeros(max(n_pairs, 1))
    P = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    args = [P(xI), P(xJ), P(start), P(wh), P(K)]
    keep = []
    if bearings is not False:   # the restatement's signature carries the bearing arrays (NULL: formed there)
        if bearings is None:
            args += [None, None]
        else:
            keep = [np.ascontiguousarray(b, np.float64) for b in bearings]
            args += [P(keep[0]), P(keep[1])]
    args += [C.c_uint64(n_pairs), C.c_double(precision), C.c_uint32(max_iterations)]
    if threads is not None:
        args.append(C.c_int(threads))
    fn.restype = C.c_double
    secs = fn(*args, P(mask), P(ok), P(F), P(prec), P(nfa))
    return dict(mask=mask[:int(start[-1])].astype(bool), ok=ok[:n_pairs].astype(bool), F=F[:n_pairs].reshape(-1, 3, 3), precision=prec[:n_pairs],
                nfa=nfa[:n_pairs], seconds=secs)


def ref_geofilter_e(tv, K, precision=4.0, max_iterations=2048, threads=0):
    """The reference's ACKernelAdaptorEssential<FivePointSolver, EpipolarDistanceError> + ACRANSAC per pair (E_ACRobust.hpp); "F" = m_E."""
    global _refgeo
    if _refgeo is None:
        _refgeo = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_geofilter.so"))
    return _geofilter_call_e(_refgeo.ref_geofilter_e_acransac, tv, K, False, precision, max_iterations, threads)


def ref_geofilter_angular(bI, bJ, start, precision_deg=4.0, max_iterations=2048, upright=False, pose_stage=False, threads=0):
    """The reference's ACKernelAdaptor_AngularRadianError<EightPointRelativePoseSolver | ThreePointUprightRelativePoseSolver, AngularError>
    + ACRANSAC per pair on bearing vectors (E_ACRobust_Angular.hpp:53-160); pose_stage: continue with RelativePoseFromEssential like
    the functor. "F" = m_E."""
    global _refgeo
    if _refgeo is None:
        _refgeo = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_geofilter.so"))
    bI = np.ascontiguousarray(bI, np.float64).reshape(-1, 3); bJ = np.ascontiguousarray(bJ, np.float64).reshape(-1, 3)
    start = np.ascontiguousarray(start, np.uint64)
    n_pairs = len(start) - 1
    mask = np.zeros(max(int(start[-1]), 1), np.uint8); ok = np.zeros(max(n_pairs, 1), np.uint8)
    F = np.zeros((max(n_pairs, 1), 9)); prec = np.zeros(max(n_pairs, 1)); nfa = np.zeros(max(n_pairs, 1))
    P = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    fn = _refgeo.ref_geofilter_e_angular_acransac
    fn.restype = C.c_double
    secs = fn(P(bI), P(bJ), P(start), C.c_uint64(n_pairs), C.c_double(precision_deg), C.c_uint32(max_iterations), C.c_int(threads), C.c_int(int(upright)),
              C.c_int(int(pose_stage)), P(mask), P(ok), P(F), P(prec), P(nfa))
    return dict(mask=mask[:int(start[-1])].astype(bool), ok=ok[:n_pairs].astype(bool), F=F[:n_pairs].reshape(-1, 3, 3), precision=prec[:n_pairs],
                nfa=nfa[:n_pairs], seconds=secs)


def ref_geofilter_eo(tv, K, precision=2.0, max_iterations=1024, threads=0):
    """The reference's ACKernelAdaptorEssentialOrtho<ThreePointKernel, OrthographicSymmetricEpipolarDistanceError> + ACRANSAC per pair as
    GeometricFilter_EOMatrix_RA::Robust_estimation runs it (Eo_Robust.hpp:50-144: pinhole cameras K, camera-plane bound); "F" = m_E."""
    global _refgeo
    if _refgeo is None:
        _refgeo = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_geofilter.so"))
    return _geofilter_call_e(_refgeo.ref_geofilter_eo_acransac, tv, K, False, precision, max_iterations, threads)


def ref_pinhole_bearings(tv, K):
    """(bI, bJ): Pinhole_Intrinsic(w, h, K)(x) of the reference for every correspondence of tv (what the essential kernel receives)"""
    global _refgeo
    if _refgeo is None:
        _refgeo = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_geofilter.so"))
    xI = np.ascontiguousarray(tv["xI"], np.float64); xJ = np.ascontiguousarray(tv["xJ"], np.float64)
    start = np.asarray(tv["start"], np.int64); K = np.ascontiguousarray(K, np.float64).reshape(-1, 18)
    bI = np.zeros((len(xI), 3)); bJ = np.zeros((len(xJ), 3))
    P = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    for p in range(len(start) - 1):
        lo, hi = int(start[p]), int(start[p + 1])
        if hi > lo:
            for x, b, k in ((xI, bI, K[p, :9]), (xJ, bJ, K[p, 9:])):
                out = np.zeros((hi - lo, 3)); xs = np.ascontiguousarray(x[lo:hi]); kk = np.ascontiguousarray(k)
                _refgeo.ref_pinhole_bearings(P(kk), P(xs), C.c_uint64(hi - lo), P(out))
                b[lo:hi] = out
    return bI, bJ


def port_geofilter_e(tv, K, precision=4.0, max_iterations=2048, bearings=None):
    """oracle/geofilter_oracle.cpp, the essential model of the restatement (its own five-point solver)."""
    return _geofilter_call_e(port().port_geofilter_e_acransac, tv, K, bearings, precision, max_iterations)


def _port_outputs(n_total, n_pairs):
    return (np.zeros(max(n_total, 1), np.uint8), np.zeros(max(n_pairs, 1), np.uint8), np.zeros((max(n_pairs, 1), 9)), np.zeros(max(n_pairs, 1)),
            np.zeros(max(n_pairs, 1)))


def port_geofilter_angular(bI, bJ, start, precision_deg=4.0, max_iterations=2048, upright=False):
    """oracle/geofilter_oracle.cpp: the a-contrario stage of the angular essential functors (its own eight-point / three-point upright solvers)"""
    bI = np.ascontiguousarray(bI, np.float64).reshape(-1, 3); bJ = np.ascontiguousarray(bJ, np.float64).reshape(-1, 3)
    start = np.ascontiguousarray(start, np.uint64)
    n_pairs = len(start) - 1
    mask, ok, F, prec, nfa = _port_outputs(int(start[-1]), n_pairs)
    P = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    fn = port().port_geofilter_e_angular_acransac
    fn.restype = C.c_double
    fn(P(bI), P(bJ), P(start), C.c_uint64(n_pairs), C.c_double(precision_deg), C.c_uint32(max_iterations), C.c_int(int(upright)), P(mask), P(ok), P(F), P(prec), P(nfa))
    return dict(mask=mask[:int(start[-1])].astype(bool), ok=ok[:n_pairs].astype(bool), F=F[:n_pairs].reshape(-1, 3, 3), precision=prec[:n_pairs], nfa=nfa[:n_pairs])


def port_geofilter_ortho(hI, hJ, start, wh, pair_bound, max_iterations=1024):
    """oracle/geofilter_oracle.cpp: the orthographic essential model on hnormalized bearing vectors with the functor's bound per pair"""
    hI = np.ascontiguousarray(hI, np.float64).reshape(-1, 2); hJ = np.ascontiguousarray(hJ, np.float64).reshape(-1, 2)
    start = np.ascontiguousarray(start, np.uint64); wh = np.ascontiguousarray(wh, np.uint32); pb = np.ascontiguousarray(pair_bound, np.float64)
    n_pairs = len(start) - 1
    mask, ok, F, prec, nfa = _port_outputs(int(start[-1]), n_pairs)
    P = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    fn = port().port_geofilter_eo_acransac
    fn.restype = C.c_double
    fn(P(hI), P(hJ), P(start), P(wh), P(pb), C.c_uint64(n_pairs), C.c_uint32(max_iterations), P(mask), P(ok), P(F), P(prec), P(nfa))
    return dict(mask=mask[:int(start[-1])].astype(bool), ok=ok[:n_pairs].astype(bool), F=F[:n_pairs].reshape(-1, 3, 3), precision=prec[:n_pairs], nfa=nfa[:n_pairs])


def port_geofilter(tv, precision=4.0, max_iterations=2048):
    """oracle/geofilter_oracle.cpp, the plain C++ restatement (one thread)."""
    return _geofilter_call(port().port_geofilter_f_acransac, tv, precision, max_iterations)


def port_geofilter_h(tv, precision=4.0, max_iterations=2048):
    """oracle/geofilter_oracle.cpp, the homography model of the restatement (Householder null vector of the 8 x 9 DLT system)."""
    return _geofilter_call(port().port_geofilter_h_acransac, tv, precision, max_iterations)


# ---- the geometric filter at container level: the same caller (oracle/ref_shim_geofilter.cpp::ref_geofilter_container) in the
# reference library and in the adapter harness (explicit specialisation of openmvg_amd/adapter/mvgx_geometric_filter.cpp) ----
ADAPTER_GEO_SO = os.path.join(ROOT, "tests", "native", "_build", "libmvgx_openmvg_adapter_geo.so")
ADAPTER_GEO_EMU_SO = os.path.join(ROOT, "tests", "native", "_build", "libmvgx_openmvg_adapter_geo_emu.so")
GEO_SINK = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32)
_geo_libs = {}


def geofilter_container_lib(kind):
    """kind: "reference" (oracle/_ref/libref_geofilter.so), "adapter" (GPU), "adapter_emu" (HIP emulation). None if unavailable."""
    if kind in _geo_libs:
        return _geo_libs[kind]
    path = {"reference": os.path.join(ROOT, "oracle", "_ref", "libref_geofilter.so"), "adapter": ADAPTER_GEO_SO, "adapter_emu": ADAPTER_GEO_EMU_SO}[kind]
    lib = None
    try:
        if kind == "adapter_emu":
            if adapter_emu() is None:   # builds the `emu` targets of the harness
                return None
        if os.path.exists(path):
            # DEEPBIND: the library's own dependencies (the emulation library / libmvgx_hip.so it was linked with) come before a
            # libmvgx_hip.so some other test loaded with RTLD_GLOBAL
            lib = C.CDLL(path, mode=os.RTLD_LOCAL | os.RTLD_NOW | os.RTLD_DEEPBIND)
            lib.ref_geofilter_container.restype = C.c_uint64
    except Exception:
        lib = None
    _geo_libs[kind] = lib
    return lib


def geofilter_container(kind, feats_xy, image_wh, putative, precision=4.0, max_iterations=2048, guided=False, ratio=0.6, k1=0.0, descs=None, model="f", focal=0.0,
                        desc_type=0):
    """feats_xy: list of (n_k, 2) float32 positions; image_wh: (n_images, 2); putative: {(I, J): (n, 2) uint32}. -> {(I, J): (m, 2)}
    desc_type: the regions the caller code builds - 0 SIFT_Regions (descs: 128 bytes per feature), 1 AKAZE_Float_Regions (64 float32),
    2 AKAZE_Binary_Regions (64 bytes)"""
    lib = geofilter_container_lib(kind)
    lib.ref_geofilter_container_region_type(C.c_int(desc_type))
    fx = np.ascontiguousarray(np.concatenate([np.asarray(f, np.float32).reshape(-1, 2) for f in feats_xy]), np.float32)
    fstart = np.cumsum([0] + [len(f) for f in feats_xy]).astype(np.uint64)
    wh = np.ascontiguousarray(image_wh, np.uint32).reshape(-1, 2)
    keys = sorted(putative)
    pij = np.ascontiguousarray(np.asarray(keys, np.uint32).reshape(-1, 2))
    mstart = np.cumsum([0] + [len(putative[k]) for k in keys]).astype(np.uint64)
    mij = np.ascontiguousarray(np.concatenate([np.asarray(putative[k], np.uint32).reshape(-1, 2) for k in keys]) if keys else np.zeros((0, 2), np.uint32))
    dd = None if descs is None else np.ascontiguousarray(np.concatenate(descs), np.float32 if desc_type == 1 else np.uint8)
    assert dd is None or dd.shape[1] == (128 if desc_type == 0 else 64)
    out = {}

    def sink(_u, I, J, p, n):
        out[(int(I), int(J))] = np.ctypeslib.as_array(p, shape=(int(n), 2)).copy() if n else np.zeros((0, 2), np.uint32)

    cb = GEO_SINK(sink)
    P = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    if model == "eo":   # the orthographic essential functor: the E container's signature
        fn = lib.ref_geofilter_container_eo
        fn.restype = C.c_uint64
        fn(P(fx), None if dd is None else P(dd), P(fstart), P(wh), C.c_uint32(len(feats_xy)), P(pij), P(mstart), P(mij), C.c_uint64(len(keys)),
           C.c_double(precision), C.c_uint32(max_iterations), C.c_int(1 if guided else 0), C.c_double(ratio), C.c_double(focal), cb, None)
        return out
    if model in ("ea", "eu"):   # the angular essential functors: (..., precision [degrees], max_iterations, upright, focal, sink, user)
        fn = lib.ref_geofilter_container_ea
        fn.restype = C.c_uint64
        fn(P(fx), None if dd is None else P(dd), P(fstart), P(wh), C.c_uint32(len(feats_xy)), P(pij), P(mstart), P(mij), C.c_uint64(len(keys)),
           C.c_double(precision), C.c_uint32(max_iterations), C.c_int(1 if model == "eu" else 0), C.c_double(focal), cb, None)
        return out
    fn = lib.ref_geofilter_container if model == "f" else lib.ref_geofilter_container_h if model == "h" else lib.ref_geofilter_container_e
    if model == "e":   # (the k1 slot of the shim carries the focal length of the views' pinhole cameras)
        k1 = focal
    fn.restype = C.c_uint64
    fn(P(fx), None if dd is None else P(dd), P(fstart), P(wh), C.c_uint32(len(feats_xy)), P(pij), P(mstart), P(mij),
                                C.c_uint64(len(keys)), C.c_double(precision), C.c_uint32(max_iterations), C.c_int(1 if guided else 0),
                                C.c_double(ratio), C.c_double(k1), cb, None)
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# guided matching (robust_estimation/guided_matching.hpp:178-227): the restatement and the reference's own template, one image pair
# ---------------------------------------------------------------------------------------------------------------------------------
def port_guided_match(kind, M, xyI, descI, xyJ, descJ, error_th, dist_ratio):
    """oracle/geofilter_oracle.cpp::port_guided_match -> (m, 2) uint32 (i, j); error_th and dist_ratio as the functors pass them (squared)"""
    L = port()
    xyI = np.ascontiguousarray(xyI, np.float64).reshape(-1, 2); xyJ = np.ascontiguousarray(xyJ, np.float64).reshape(-1, 2)
    descI = np.ascontiguousarray(descI, np.uint8); descJ = np.ascontiguousarray(descJ, np.uint8)
    nb = descI.shape[1] if descI.ndim == 2 and len(descI) else (descJ.shape[1] if descJ.ndim == 2 and len(descJ) else 128)
    M = np.ascontiguousarray(M, np.float64).reshape(9)
    out = np.zeros((max(len(xyI), 1), 2), np.uint32)
    L.port_guided_match.restype = C.c_uint64
    n = L.port_guided_match(C.c_int(kind), M.ctypes.data_as(C.c_void_p), xyI.ctypes.data_as(C.c_void_p), descI.ctypes.data_as(C.c_void_p), C.c_uint64(len(xyI)),
                            xyJ.ctypes.data_as(C.c_void_p), descJ.ctypes.data_as(C.c_void_p), C.c_uint64(len(xyJ)), C.c_uint32(nb), C.c_double(error_th),
                            C.c_double(dist_ratio), out.ctypes.data_as(C.c_void_p))
    return out[:int(n)].copy()


def _typed_desc(desc, desc_type):
    """(rows as the C side reads them, elements per row) for desc_type 0 uint8 / 1 float32 / 2 bit rows"""
    a = np.ascontiguousarray(desc, np.float32 if desc_type == 1 else np.uint8)
    return a, (a.shape[1] if a.ndim == 2 else 0)


def port_guided_match_typed(kind, desc_type, M, xyI, descI, xyJ, descJ, error_th, dist_ratio):
    """oracle/geofilter_oracle.cpp::port_guided_match_typed (desc_type 0 uint8 L2, 1 float L2, 2 squared Hamming) -> (m, 2) uint32"""
    L = port()
    xyI = np.ascontiguousarray(xyI, np.float64).reshape(-1, 2); xyJ = np.ascontiguousarray(xyJ, np.float64).reshape(-1, 2)
    dI, nI_ = _typed_desc(descI, desc_type); dJ, nJ_ = _typed_desc(descJ, desc_type)
    n_el = nI_ or nJ_ or 64
    M = np.ascontiguousarray(M, np.float64).reshape(9)
    out = np.zeros((max(len(xyI), 1), 2), np.uint32)
    L.port_guided_match_typed.restype = C.c_uint64
    n = L.port_guided_match_typed(C.c_int(kind), C.c_int(desc_type), M.ctypes.data_as(C.c_void_p), xyI.ctypes.data_as(C.c_void_p), dI.ctypes.data_as(C.c_void_p),
                                  C.c_uint64(len(xyI)), xyJ.ctypes.data_as(C.c_void_p), dJ.ctypes.data_as(C.c_void_p), C.c_uint64(len(xyJ)), C.c_uint32(n_el),
                                  C.c_double(error_th), C.c_double(dist_ratio), out.ctypes.data_as(C.c_void_p))
    return out[:int(n)].copy()


def ref_guided_match_typed(kind, desc_type, M, xyI, descI, xyJ, descJ, error_th, dist_ratio):
    """the reference's GuidedMatching template on AKAZE_Float_Regions (desc_type 1: 64 floats per row) / AKAZE_Binary_Regions (2: 64 bytes)
    / SIFT_Regions (0) built from the arrays -> (m, 2) uint32"""
    global _refgeo
    if _refgeo is None:
        _refgeo = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_geofilter.so"))
    xI = np.ascontiguousarray(xyI, np.float32).reshape(-1, 2); xJ = np.ascontiguousarray(xyJ, np.float32).reshape(-1, 2)
    assert np.array_equal(xI.astype(np.float64), np.asarray(xyI, np.float64).reshape(-1, 2)) and np.array_equal(xJ.astype(np.float64), np.asarray(xyJ, np.float64).reshape(-1, 2))
    dI, _ = _typed_desc(descI, desc_type); dJ, _ = _typed_desc(descJ, desc_type)
    want_len = {0: 128, 1: 64, 2: 64}[desc_type]
    assert (not len(dI) or dI.shape[1] == want_len) and (not len(dJ) or dJ.shape[1] == want_len)
    M = np.ascontiguousarray(M, np.float64).reshape(9)
    out = np.zeros((max(len(xI), 1), 2), np.uint32)
    _refgeo.ref_guided_match_typed.restype = C.c_uint64
    n = _refgeo.ref_guided_match_typed(C.c_int(kind), C.c_int(desc_type), M.ctypes.data_as(C.c_void_p), xI.ctypes.data_as(C.c_void_p), dI.ctypes.data_as(C.c_void_p),
                                       C.c_uint64(len(xI)), xJ.ctypes.data_as(C.c_void_p), dJ.ctypes.data_as(C.c_void_p), C.c_uint64(len(xJ)), C.c_double(error_th),
                                       C.c_double(dist_ratio), out.ctypes.data_as(C.c_void_p))
    return out[:int(n)].copy()


def ref_guided_match(kind, M, xyI, descI, xyJ, descJ, error_th, dist_ratio):
    """the reference's GuidedMatching<Mat3, EpipolarDistanceError | AsymmetricError> on SIFT_Regions built from the arrays (128-byte
    descriptors; positions must be exact in float: the regions store floats) -> (m, 2) uint32"""
    global _refgeo
    if _refgeo is None:
        _refgeo = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_geofilter.so"))
    xI = np.ascontiguousarray(xyI, np.float32).reshape(-1, 2); xJ = np.ascontiguousarray(xyJ, np.float32).reshape(-1, 2)
    assert np.array_equal(xI.astype(np.float64), np.asarray(xyI, np.float64).reshape(-1, 2)) and np.array_equal(xJ.astype(np.float64), np.asarray(xyJ, np.float64).reshape(-1, 2))
    descI = np.ascontiguousarray(descI, np.uint8).reshape(-1, 128); descJ = np.ascontiguousarray(descJ, np.uint8).reshape(-1, 128)
    M = np.ascontiguousarray(M, np.float64).reshape(9)
    out = np.zeros((max(len(xI), 1), 2), np.uint32)
    _refgeo.ref_guided_match.restype = C.c_uint64
    n = _refgeo.ref_guided_match(C.c_int(kind), M.ctypes.data_as(C.c_void_p), xI.ctypes.data_as(C.c_void_p), descI.ctypes.data_as(C.c_void_p), C.c_uint64(len(xI)),
                                 xJ.ctypes.data_as(C.c_void_p), descJ.ctypes.data_as(C.c_void_p), C.c_uint64(len(xJ)), C.c_double(error_th), C.c_double(dist_ratio),
                                 out.ctypes.data_as(C.c_void_p))
    return out[:int(n)].copy()
