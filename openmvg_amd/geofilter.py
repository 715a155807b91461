"""Host-side mirror of openMVG's geometric filtering of putative matches on top of the mvgx C ABI.

  ImageCollectionGeometricFilter::Robust_model_estimation      matching_image_collection/GeometricFilter.hpp:66-131
  GeometricFilter_FMatrix_AC(dPrecision, iteration)             matching_image_collection/F_ACRobust.hpp:32-122
  GeometricFilter_HMatrix_AC(dPrecision, iteration)             matching_image_collection/H_ACRobust.hpp:32-113
  MatchesPairToMat                                              matching_image_collection/Geometric_Filter_utils.hpp:56-64

All numerics run in libmvgx_hip.so on the GPU (one wave per image pair); there is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _capi


class GeometricFilter_FMatrix_AC:
    """Field names of the reference functor; `Robust_estimation` of one pair is `filter_pairs` on a one-pair container."""

    _entry, _entry_indexed = "mvgx_geofilter_f_acransac", "mvgx_geofilter_f_acransac_indexed"

    def __init__(self, dPrecision=4.0, iteration=1024):
        self.m_dPrecision = float(dPrecision)
        self.m_stIteration = int(iteration)


class GeometricFilter_HMatrix_AC(GeometricFilter_FMatrix_AC):
    """The homography functor (H_ACRobust.hpp:32-113): same fields; the result's "F" field then holds m_H (x_J ~ H x_I)."""
    _entry, _entry_indexed = "mvgx_geofilter_h_acransac", "mvgx_geofilter_h_acransac_indexed"


class GeometricFilter_EMatrix_AC(GeometricFilter_FMatrix_AC):
    """The essential-matrix functor (E_ACRobust.hpp:39-150): same fields; the result's "F" field holds m_E, "precision_robust" the
    squared pixel bound the reference stores. Needs the calibration matrices (filter_pairs_e)."""
    _entry, _entry_indexed = "mvgx_geofilter_e_acransac", "mvgx_geofilter_e_acransac_indexed"


class GeometricFilter_ESphericalMatrix_AC_Angular(GeometricFilter_FMatrix_AC):
    """The angular essential functor (E_ACRobust_Angular.hpp:33-191): isUprightEssentialMatrix selects the three-point upright solver
    instead of the eight-point one; m_dPrecision is the precision_upper_bound in DEGREES. The result's "F" field holds m_E (unit
    norm), "precision_robust" an angle in radians. filter_pairs_angular runs the a-contrario stage; the functor's cheirality
    stage (RelativePoseFromEssential) is the caller's."""

    def __init__(self, precision_upper_bound=4.0, iteration=1024, isUprightEssentialMatrix=False):
        super().__init__(precision_upper_bound, iteration)
        self.isUprightEssentialMatrix = bool(isUprightEssentialMatrix)


def filter_pairs_angular(bI, bJ, match_start, functor=None, device=-1):
    """bI, bJ: (N, 3) bearing vectors of the putative matches of all pairs (what the cameras' operator() returns for the matched
    positions), pair p owning rows [match_start[p], match_start[p + 1]). Returns (inlier_mask, results, stats) like filter_pairs."""
    functor = functor or GeometricFilter_ESphericalMatrix_AC_Angular(4.0, 2048)
    bI = np.ascontiguousarray(bI, np.float64).reshape(-1, 3)
    bJ = np.ascontiguousarray(bJ, np.float64).reshape(-1, 3)
    start = np.ascontiguousarray(match_start, np.uint64)
    n_pairs = len(start) - 1
    if int(start[-1]) != len(bI) or len(bI) != len(bJ):
        raise ValueError("filter_pairs_angular: inconsistent array sizes")
    mask = np.zeros(max(len(bI), 1), np.uint8)
    res = (_capi.GeofilterResult * max(n_pairs, 1))()
    st = _capi.GeofilterStats()
    opt = _capi.GeofilterOptions(functor.m_dPrecision, functor.m_stIteration)
    P = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    _capi.check(_capi.lib().mvgx_geofilter_e_angular_acransac(int(device), P(bI), P(bJ), P(start), n_pairs, int(functor.isUprightEssentialMatrix),
                                                              C.byref(opt), P(mask), C.cast(res, C.c_void_p), C.byref(st)))
    return mask[:len(bI)].astype(bool), _results_array(res, n_pairs), st


class GeometricFilter_EOMatrix_RA(GeometricFilter_FMatrix_AC):
    """The orthographic essential functor (Eo_Robust.hpp:35-165): same fields; m_dPrecision in pixels - filter_pairs_ortho turns it
    into the camera-plane bound of every pair like the functor (mean of precision^2 / focal of the two cameras)."""


def filter_pairs_ortho(xI, xJ, match_start, image_wh, K, functor=None, device=-1):
    """The orthographic essential model on gathered correspondences (pixels xI / xJ of pinhole cameras K (n_pairs, 2, 3, 3)): the
    hnormalized bearing vectors and the per-pair bound are formed here the way Eo_Robust.hpp:90-121 forms them.
    Returns (inlier_mask, results, stats) like filter_pairs."""
    functor = functor or GeometricFilter_EOMatrix_RA(2.0, 1024)
    xI = np.ascontiguousarray(xI, np.float64).reshape(-1, 2)
    xJ = np.ascontiguousarray(xJ, np.float64).reshape(-1, 2)
    start = np.ascontiguousarray(match_start, np.uint64)
    wh = np.ascontiguousarray(image_wh, np.uint32).reshape(-1, 4)
    n_pairs = len(start) - 1
    K = np.ascontiguousarray(K, np.float64).reshape(-1, 2, 3, 3)
    if len(wh) != n_pairs or len(K) != n_pairs or int(start[-1]) != len(xI) or len(xI) != len(xJ):
        raise ValueError("filter_pairs_ortho: inconsistent array sizes")
    hI, hJ, prec = ortho_inputs(xI, xJ, start, K, functor.m_dPrecision)
    return filter_pairs_ortho_prepared(hI, hJ, start, wh, prec, functor, device)


def filter_pairs_ortho_prepared(hI, hJ, match_start, image_wh, pair_precision, functor=None, device=-1):
    """mvgx_geofilter_eo_acransac on inputs in the entry's own terms: hnormalized bearing vectors (N, 2) and the bound of every pair"""
    functor = functor or GeometricFilter_EOMatrix_RA(2.0, 1024)
    hI = np.ascontiguousarray(hI, np.float64).reshape(-1, 2); hJ = np.ascontiguousarray(hJ, np.float64).reshape(-1, 2)
    start = np.ascontiguousarray(match_start, np.uint64); wh = np.ascontiguousarray(image_wh, np.uint32).reshape(-1, 4)
    prec = np.ascontiguousarray(pair_precision, np.float64)
    n_pairs = len(start) - 1
    if len(wh) != n_pairs or len(prec) < n_pairs or int(start[-1]) != len(hI) or len(hI) != len(hJ):
        raise ValueError("filter_pairs_ortho_prepared: inconsistent array sizes")
    mask = np.zeros(max(len(hI), 1), np.uint8)
    res = (_capi.GeofilterResult * max(n_pairs, 1))()
    st = _capi.GeofilterStats()
    opt = _capi.GeofilterOptions(functor.m_dPrecision, functor.m_stIteration)
    P = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    _capi.check(_capi.lib().mvgx_geofilter_eo_acransac(int(device), P(hI), P(hJ), P(start), P(wh), P(prec), n_pairs, C.byref(opt), P(mask),
                                                       C.cast(res, C.c_void_p), C.byref(st)))
    return mask[:len(hI)].astype(bool), _results_array(res, n_pairs), st


def ortho_inputs(xI, xJ, start, K, precision):
    """(hI, hJ, pair_precision): hnormalized pinhole bearings of the correspondences and (precision^2 / f_I + precision^2 / f_J) / 2 per pair
    (Pinhole_Intrinsic::imagePlane_toCameraPlaneError, Camera_Pinhole.hpp:195-198)"""
    st = np.asarray(start, np.int64)
    hI = np.zeros((len(xI), 2)); hJ = np.zeros((len(xJ), 2)); prec = np.zeros(max(len(st) - 1, 1))
    for p in range(len(st) - 1):
        lo, hi = int(st[p]), int(st[p + 1])
        for x, h, k in ((xI, hI, K[p, 0]), (xJ, hJ, K[p, 1])):
            if hi > lo:
                b = pinhole_bearings(k, x[lo:hi])
                h[lo:hi] = b[:, :2] / b[:, 2:3]
        prec[p] = (precision * precision / K[p, 0, 0, 0] + precision * precision / K[p, 1, 0, 0]) / 2.0
    return np.ascontiguousarray(hI), np.ascontiguousarray(hJ), np.ascontiguousarray(prec)


def pinhole_bearings(K, x):
    """Pinhole_Intrinsic::operator()(x) (Camera_Pinhole.hpp:136-139): normalised Kinv (x, y, 1) per point; K (3, 3), x (n, 2) -> (n, 3).
    (Host mirror for callers without the camera class at hand; the openMVG adapter calls the camera's own operator.)"""
    K = np.asarray(K, np.float64).reshape(3, 3)
    x = np.asarray(x, np.float64).reshape(-1, 2)
    v = np.concatenate([x, np.ones((len(x), 1))], 1) @ np.linalg.inv(K).T
    return np.ascontiguousarray(v / np.linalg.norm(v, axis=1, keepdims=True))


def filter_pairs_e(xI, xJ, match_start, image_wh, K, functor=None, device=-1, bearings=None):
    """The essential model on gathered correspondences: K (n_pairs, 2, 3, 3) = {K_I, K_J} per pair; bearings = (bI, bJ), (N, 3) each -
    what the cameras' operator() returns for xI / xJ - or None: pinhole_bearings. Returns (inlier_mask, results, stats) like filter_pairs."""
    functor = functor or GeometricFilter_EMatrix_AC(4.0, 2048)
    xI = np.ascontiguousarray(xI, np.float64).reshape(-1, 2)
    xJ = np.ascontiguousarray(xJ, np.float64).reshape(-1, 2)
    start = np.ascontiguousarray(match_start, np.uint64)
    wh = np.ascontiguousarray(image_wh, np.uint32).reshape(-1, 4)
    n_pairs = len(start) - 1
    K = np.ascontiguousarray(K, np.float64).reshape(-1, 18)
    if len(wh) != n_pairs or len(K) != n_pairs or int(start[-1]) != len(xI) or len(xI) != len(xJ):
        raise ValueError("filter_pairs_e: inconsistent array sizes")
    if bearings is None:
        bI = np.zeros((len(xI), 3)); bJ = np.zeros((len(xJ), 3))
        for p in range(n_pairs):
            lo, hi = int(start[p]), int(start[p + 1])
            bI[lo:hi] = pinhole_bearings(K[p, :9], xI[lo:hi]); bJ[lo:hi] = pinhole_bearings(K[p, 9:], xJ[lo:hi])
    else:
        bI, bJ = (np.ascontiguousarray(b, np.float64).reshape(-1, 3) for b in bearings)
    bI = np.ascontiguousarray(bI); bJ = np.ascontiguousarray(bJ)
    mask = np.zeros(max(len(xI), 1), np.uint8)
    res = (_capi.GeofilterResult * max(n_pairs, 1))()
    st = _capi.GeofilterStats()
    opt = _capi.GeofilterOptions(functor.m_dPrecision, functor.m_stIteration)
    P = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    _capi.check(_capi.lib().mvgx_geofilter_e_acransac(int(device), P(xI), P(xJ), P(bI), P(bJ), P(start), P(wh), P(K), n_pairs, C.byref(opt), P(mask),
                                                      C.cast(res, C.c_void_p), C.byref(st)))
    return mask[:len(xI)].astype(bool), _results_array(res, n_pairs), st


def filter_pairs(xI, xJ, match_start, image_wh, functor=None, device=-1):
    """xI, xJ: (N, 2) float64 pixel positions of the putative matches of all pairs, pair p owning rows
    [match_start[p], match_start[p + 1]); image_wh: (n_pairs, 4) uint32 {w_I, h_I, w_J, h_J}.
    Returns (inlier_mask (N,) bool, results structured array, stats)."""
    functor = functor or GeometricFilter_FMatrix_AC(4.0, 2048)
    xI = np.ascontiguousarray(xI, np.float64).reshape(-1, 2)
    xJ = np.ascontiguousarray(xJ, np.float64).reshape(-1, 2)
    start = np.ascontiguousarray(match_start, np.uint64)
    wh = np.ascontiguousarray(image_wh, np.uint32).reshape(-1, 4)
    n_pairs = len(start) - 1
    if len(wh) != n_pairs or int(start[-1]) != len(xI) or len(xI) != len(xJ):
        raise ValueError("filter_pairs: inconsistent array sizes")
    mask = np.zeros(max(len(xI), 1), np.uint8)
    res = (_capi.GeofilterResult * max(n_pairs, 1))()
    st = _capi.GeofilterStats()
    opt = _capi.GeofilterOptions(functor.m_dPrecision, functor.m_stIteration)
    P = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    _capi.check(getattr(_capi.lib(), functor._entry)(int(device), P(xI), P(xJ), P(start), P(wh), n_pairs, C.byref(opt), P(mask),
                                                      C.cast(res, C.c_void_p), C.byref(st)))
    out = _results_array(res, n_pairs)
    return mask[:len(xI)].astype(bool), out, st


def _results_array(res, n_pairs):
    out = np.zeros(n_pairs, dtype=[("F", np.float64, (3, 3)), ("precision_robust", np.float64), ("nfa", np.float64),
                                   ("n_inliers", np.uint32), ("ok", bool)])
    if n_pairs:
        raw = np.frombuffer(res, dtype=np.dtype([("F", np.float64, (9,)), ("precision_robust", np.float64), ("nfa", np.float64),
                                                 ("n_inliers", np.uint32), ("ok", np.int32)], align=True), count=n_pairs)
        out["F"] = raw["F"].reshape(-1, 3, 3); out["precision_robust"] = raw["precision_robust"]; out["nfa"] = raw["nfa"]
        out["n_inliers"] = raw["n_inliers"]; out["ok"] = raw["ok"] != 0
    return out


def filter_pairs_indexed(feats_xy, image_sizes, pairs, match_start, ij, functor=None, device=-1):
    """The PairWiseMatches-shaped form (mvgx_geofilter_f_acransac_indexed): feats_xy = list of (n_k, 2) feature positions per image
    (or one (N, 2) array with feat_start), image_sizes (n_images, 2) {w, h}, pairs (n_pairs, 2) image ids, ij (n_matches, 2) index
    pairs with pair p owning rows [match_start[p], match_start[p + 1]). The positions are gathered on the device.
    Returns (inlier_mask (n_matches,) bool, results structured array, stats)."""
    functor = functor or GeometricFilter_FMatrix_AC(4.0, 2048)
    if isinstance(feats_xy, tuple):
        feat, fstart = feats_xy
        feat = np.ascontiguousarray(feat, np.float64).reshape(-1, 2); fstart = np.ascontiguousarray(fstart, np.uint64)
    else:
        counts = [len(f) for f in feats_xy]
        fstart = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
        feat = np.ascontiguousarray(np.concatenate([np.asarray(f, np.float64).reshape(-1, 2) for f in feats_xy]) if counts else np.zeros((0, 2)))
    n_images = len(fstart) - 1
    wh = np.ascontiguousarray(image_sizes, np.uint32).reshape(-1, 2)
    pairs = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
    start = np.ascontiguousarray(match_start, np.uint64)
    ij = np.ascontiguousarray(ij, np.uint32).reshape(-1, 2)
    n_pairs = len(pairs)
    if len(wh) != n_images or len(start) != n_pairs + 1 or int(start[-1]) != len(ij):
        raise ValueError("filter_pairs_indexed: inconsistent array sizes")
    mask = np.zeros(max(len(ij), 1), np.uint8)
    res = (_capi.GeofilterResult * max(n_pairs, 1))()
    st = _capi.GeofilterStats()
    opt = _capi.GeofilterOptions(functor.m_dPrecision, functor.m_stIteration)
    P = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    _capi.check(getattr(_capi.lib(), functor._entry_indexed)(int(device), P(feat), P(fstart), P(wh), n_images, P(pairs), P(start), P(ij), n_pairs,
                                                              C.byref(opt), P(mask), C.cast(res, C.c_void_p), C.byref(st)))
    return mask[:len(ij)].astype(bool), _results_array(res, n_pairs), st


def Robust_model_estimation(putative_matches, feats_xy, image_sizes, functor=None, device=-1):
    """The container form: putative_matches = {(I, J): (n, 2) uint32 index pairs} (a PairWiseMatches), feats_xy[k] = (n_k, 2)
    feature positions of image k (undistorted where the reference would undistort them), image_sizes[k] = (w, h). Returns the
    geometric matches {(I, J): (m, 2)}: only the pairs whose estimation succeeded, like _map_GeometricMatches."""
    keys = sorted(putative_matches)
    if not keys:
        return {}
    n_images = len(feats_xy)
    lists = [np.asarray(putative_matches[k], np.uint32).reshape(-1, 2) for k in keys]
    start = np.concatenate([[0], np.cumsum([len(m) for m in lists])]).astype(np.uint64)
    sizes = np.array([image_sizes[k] for k in range(n_images)], np.uint32).reshape(-1, 2)
    mask, res, _ = filter_pairs_indexed([feats_xy[k] for k in range(n_images)], sizes, np.array(keys, np.uint32),
                                        start, np.concatenate(lists) if len(lists) else np.zeros((0, 2), np.uint32), functor, device)
    out = {}
    for k, key in enumerate(keys):
        if res["ok"][k]:
            out[key] = np.asarray(putative_matches[key], np.uint32).reshape(-1, 2)[mask[start[k]:start[k + 1]]]
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# Guided matching: robust_estimation/guided_matching.hpp:178-227 through {F,H,E}_ACRobust.hpp's Geometry_guided_matching
# ---------------------------------------------------------------------------------------------------------------------------------
GUIDED_FUNDAMENTAL, GUIDED_HOMOGRAPHY = 0, 1
DESC_U8, DESC_F32, DESC_BINARY = 0, 1, 2


def guided_matching(feat_xy, descs, pairs, models, precision_robust, dDistanceRatio=0.6, kind=GUIDED_FUNDAMENTAL, device=-1, desc_type=None):
    """The functors' second stage for a list of image pairs on the device (mvgx_guided_match).

    feat_xy: per image an (n, 2) array of the positions the reference compares (cam->get_ud_pixel(position), or the position itself);
    descs: per image an (n, 64 | 128 | 144) uint8 array (SIFT / LIOP: L2<uint8_t>), an (n, 64 | 128) float32 array (AKAZE float: L2<float>)
    or, with desc_type=DESC_BINARY, an (n, 32 | 64) uint8 array of bit rows (AKAZE binary: squared Hamming distance); desc_type None:
    by dtype (float32 -> DESC_F32, else DESC_U8); pairs: (P, 2) image indices; models: (P, 3, 3) - m_F (for the essential
    functor F = K2^-T E K1^-1) or m_H; precision_robust: (P,) m_dPrecision_robust in pixels (infinity: no guided matching for that
    pair, as the reference). The reference passes Square(m_dPrecision_robust) and Square(dDistanceRatio): so does this function.
    Returns ({(I, J): (m, 2) uint32 array of (i, j)} for the pairs with at least one match, stats)."""
    n_images = len(feat_xy)
    start = np.zeros(n_images + 1, np.uint64)
    for k in range(n_images):
        if len(feat_xy[k]) != len(descs[k]):
            raise ValueError("guided_matching: positions and descriptors of an image differ in length")
        start[k + 1] = start[k] + len(feat_xy[k])
    xy = np.ascontiguousarray(np.concatenate([np.asarray(f, np.float64).reshape(-1, 2) for f in feat_xy]) if n_images else np.zeros((0, 2)))
    nb = {int(np.asarray(d).shape[1]) for d in descs if len(d)} or {128}
    if len(nb) != 1:
        raise ValueError("guided_matching: descriptors of different lengths")
    desc_len = nb.pop()
    if desc_type is None:
        desc_type = DESC_F32 if any(np.asarray(d).dtype == np.float32 for d in descs if len(d)) else DESC_U8
    dt = np.float32 if desc_type == DESC_F32 else np.uint8
    desc_bytes = desc_len * np.dtype(dt).itemsize
    dd = np.ascontiguousarray(np.concatenate([np.asarray(d, dt).reshape(-1, desc_len) for d in descs]) if n_images else np.zeros((0, desc_len), dt))
    pairs = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
    models = np.ascontiguousarray(models, np.float64).reshape(-1, 9)
    prec = np.asarray(precision_robust, np.float64).reshape(-1)
    if not (len(pairs) == len(models) == len(prec)):
        raise ValueError("guided_matching: one model and one precision per pair")
    with np.errstate(over="ignore"):
        th = np.ascontiguousarray(prec * prec)   # Square(m_dPrecision_robust)
    ratio_sq = float(dDistanceRatio) * float(dDistanceRatio)
    ms = np.zeros(len(pairs) + 1, np.uint64)
    out = C.c_void_p()
    st = _capi.GuidedStats()
    _capi.check(_capi.lib().mvgx_guided_match(device, xy.ctypes.data, dd.ctypes.data, int(desc_type), desc_bytes, start.ctypes.data, n_images, pairs.ctypes.data,
                                              models.ctypes.data, th.ctypes.data, len(pairs), int(kind), ratio_sq, ms.ctypes.data, C.byref(out), C.byref(st)))
    try:
        total = int(ms[-1])
        if total and out:   # (no pair / no match: the library hands back NULL - as_array on it raises)
            ij = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint32)), shape=(total * 2,)).reshape(-1, 2).copy()
        else:
            ij = np.zeros((0, 2), np.uint32)
    finally:
        _capi.lib().mvgx_host_free(out)   # (free(NULL) is fine)
    res = {}
    for p in range(len(pairs)):
        lo, hi = int(ms[p]), int(ms[p + 1])
        if hi > lo:
            res[(int(pairs[p, 0]), int(pairs[p, 1]))] = ij[lo:hi]
    return res, st
