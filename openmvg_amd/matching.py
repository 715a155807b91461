"""Host-side mirror of openMVG's collection-matching interface on top of the mvgx C ABI.

Names, argument meaning and skip/insert behaviour follow the reference (paths under /root/reference/src/openMVG):

  EMatcherType                         matching/matcher_type.hpp:15-24
  exhaustivePairs / contiguousWithOverlap   matching_image_collection/Pair_Builder.hpp:25-44
  Regions (Scalar_Regions<.., uint8, 128>)  features/scalar_regions.hpp:28-138, features/regions_factory.hpp:19
  Regions_Provider.get                 sfm/pipelines/sfm_regions_provider.hpp:76-85
  PairWiseMatches                      matching/indMatch.hpp:70-96  (map Pair -> IndMatches)
  Matcher_Regions(distRatio, type).Match(provider, pairs, map_PutativeMatches, progress)
                                       matching_image_collection/Matcher_Regions.{hpp:28-51,cpp:22-107}

Only the path the MI355X kernel accelerates is implemented: BRUTE_FORCE_L2 on uint8 x 128 descriptors.
Anything else raises (there is no CPU fallback in this package; unchanged openMVG keeps its own CPU matchers).
"""
import ctypes as C
from enum import IntEnum

import numpy as np

from . import _capi


class EMatcherType(IntEnum):
    BRUTE_FORCE_L2 = 0
    ANN_L2 = 1
    CASCADE_HASHING_L2 = 2
    HNSW_L2 = 3
    HNSW_L1 = 4
    BRUTE_FORCE_HAMMING = 5
    HNSW_HAMMING = 6


def exhaustivePairs(N):
    """All (I, J), I < J, in the sorted order of a std::set<Pair> (Pair_Builder.hpp:25-33)."""
    return [(i, j) for i in range(N) for j in range(i + 1, N)]


def contiguousWithOverlap(N, overlapSize):
    """Pair_Builder.hpp:37-44."""
    return [(i, j) for i in range(N) for j in range(i + 1, min(i + 1 + overlapSize, N))]


def exhaustive_pairs_array(N):
    """exhaustivePairs as an (n_pairs, 2) uint32 array, same order, without Python tuples (10 000 images: 5e7 rows, ~1 s)."""
    N = int(N)
    if N < 2:
        return np.zeros((0, 2), np.uint32)
    out = np.empty((N * (N - 1) // 2, 2), np.uint32)
    js = np.arange(N, dtype=np.uint32)
    pos = 0
    for i in range(N - 1):           # N slice assignments: memory-bound
        n = N - 1 - i
        out[pos:pos + n, 0] = i
        out[pos:pos + n, 1] = js[i + 1:]
        pos += n
    return out


class Regions:
    """SIFT_Regions stand-in: an (n, 128) uint8 row-major descriptor array (DescriptorRawData layout)."""

    def __init__(self, descriptors):
        d = np.ascontiguousarray(descriptors, dtype=np.uint8)
        if d.ndim != 2:
            raise ValueError("descriptors must be a 2-D array (n, L)")
        self._d = d

    def RegionCount(self):
        return int(self._d.shape[0])

    def DescriptorLength(self):
        return int(self._d.shape[1])

    def Type_id(self):
        return "h"  # typeid(unsigned char).name() under the Itanium ABI

    def IsScalar(self):
        return True

    def IsBinary(self):
        return False

    def DescriptorRawData(self):
        return self._d


class Binary_Regions(Regions):
    """Binary_Regions<SIOPointFeature, L> stand-in (features/binary_regions.hpp; AKAZE_Binary_Regions: L = 64): an (n, L)
    uint8 array of packed bits."""

    def IsScalar(self):
        return False

    def IsBinary(self):
        return True


class Regions_Provider:
    """id_view -> Regions cache, fully loaded up-front like the reference provider."""

    def __init__(self, regions_by_view=None):
        self.cache_ = dict(regions_by_view or {})

    def get(self, x):
        return self.cache_.get(x)


class PairWiseMatches(dict):
    """Pair -> (n, 2) uint32 array of IndMatch(i_, j_)."""

    def insert(self, pair, ind_matches):
        self.setdefault(pair, ind_matches)  # std::map::insert keeps an existing entry


class MatchContext:
    """Device-resident descriptor set + runs over pair lists (thin wrapper over mvgx_match_*)."""

    def __init__(self, device=-1, devices=None):
        """device: one ordinal (-1: MVGX_DEVICES or the current device). devices: a list of ordinals -> one context over
        several devices of this process (mvgx_match_create_multi; an ordinal may repeat)."""
        self._h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int * len(devices))(*[int(d) for d in devices])
            _capi.check(_capi.lib().mvgx_match_create_multi(arr, len(devices), C.byref(self._h)))
        else:
            _capi.check(_capi.lib().mvgx_match_create(int(device), C.byref(self._h)))
        self.n_images = 0
        self._keep = None

    def close(self):
        if self._h:
            _capi.lib().mvgx_match_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        _capi.check(_capi.lib().mvgx_match_set_option(self._h, key.encode(), int(value)))

    def set_regions(self, desc_list):
        """desc_list: sequence of (n_k, 128) uint8 arrays (n_k may be 0)."""
        arrs = []
        dim = 128
        for d in desc_list:
            a = np.ascontiguousarray(d, dtype=np.uint8)
            if a.size == 0:
                a = np.zeros((0, 128), np.uint8)
            if a.ndim != 2:
                raise ValueError("each descriptor array must be 2-D (n, L)")
            if a.shape[0] and a.shape[1] != 128:
                dim = a.shape[1]  # rejected by the C side (MVGX_ERR_UNSUPPORTED)
            arrs.append(a)
        n = len(arrs)
        ptrs = (C.c_void_p * max(n, 1))()
        cnt = (C.c_uint32 * max(n, 1))()
        for k, a in enumerate(arrs):
            ptrs[k] = a.ctypes.data if a.shape[0] else None
            cnt[k] = a.shape[0]
        self._keep = arrs
        _capi.check(_capi.lib().mvgx_match_set_regions(self._h, ptrs, cnt, n, dim))
        self.n_images = n

    def set_regions_device(self, d_ptr, n_desc):
        n_desc = np.ascontiguousarray(n_desc, dtype=np.uint32)
        _capi.check(_capi.lib().mvgx_match_set_regions_device(
            self._h, C.c_void_p(int(d_ptr)), n_desc.ctypes.data_as(C.POINTER(C.c_uint32)), len(n_desc), 128))
        self.n_images = len(n_desc)

    def run(self, pairs, ratio_sq, fetch=True):
        """pairs: (n_pairs, 2) uint32. Returns (stats, offsets[n_pairs+1] uint64, ij[(n_matches, 2)] uint32)."""
        pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        st = _capi.MatchStats()
        _capi.check(_capi.lib().mvgx_match_run(self._h, pairs.ctypes.data, pairs.shape[0], np.float32(ratio_sq), C.byref(st)))
        if not fetch:
            return st, None, None
        po = C.POINTER(C.c_uint64)()
        pij = C.POINTER(C.c_uint32)()
        _capi.check(_capi.lib().mvgx_match_results(self._h, C.byref(po), C.byref(pij)))
        n = pairs.shape[0]
        offsets = np.ctypeslib.as_array(po, shape=(n + 1,)).copy() if n + 1 > 0 else np.zeros(1, np.uint64)
        total = int(offsets[-1])
        ij = np.ctypeslib.as_array(pij, shape=(total, 2)).copy() if total else np.zeros((0, 2), np.uint32)
        return st, offsets, ij


    def run_stream(self, pairs, ratio_sq, on_batch=None):
        """mvgx_match_run_stream: `on_batch(first_pair, offsets[nb + 1] uint32, ij[(n, 2)] uint32)` is called on this thread
        for every batch (arrays are views valid during the call; return a true value to stop). Returns the stats."""
        pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        st = _capi.MatchStats()
        err = []

        def sink(_user, first_pair, nb, offsets, ij):
            if on_batch is None:
                return 0
            try:
                off = np.ctypeslib.as_array(offsets, shape=(nb + 1,))
                total = int(off[nb])
                lists = np.ctypeslib.as_array(ij, shape=(total, 2)) if total else np.zeros((0, 2), np.uint32)
                return 1 if on_batch(int(first_pair), off, lists) else 0
            except BaseException as e:   # never unwind through the C frames
                err.append(e)
                return 1

        cb = _capi.MATCH_BATCH_SINK(sink)
        _capi.check(_capi.lib().mvgx_match_run_stream(self._h, pairs.ctypes.data, pairs.shape[0], np.float32(ratio_sq), cb, None,
                                                      C.byref(st)))
        if err:
            raise err[0]
        return st

    def run_collect_stream(self, pairs, ratio_sq):
        """run_stream assembled into the (stats, offsets, ij) shape of run(): for tests of the streaming path."""
        pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        counts = np.zeros(len(pairs) + 1, np.uint64)
        pieces = {}

        def on_batch(p0, off, lists):
            counts[p0 + 1:p0 + len(off)] = np.diff(off.astype(np.int64)).astype(np.uint64)
            pieces[p0] = lists.copy()

        st = self.run_stream(pairs, ratio_sq, on_batch)
        offsets = np.cumsum(counts, dtype=np.uint64)
        keys = sorted(pieces)
        ij = np.concatenate([pieces[k] for k in keys]) if keys else np.zeros((0, 2), np.uint32)
        return st, offsets, ij, keys


class HammingContext:
    """Device-resident binary descriptor set + runs over pair lists (thin wrapper over mvgx_hamming_*)."""
    _prefix = "mvgx_hamming"
    _dtype = np.uint8

    def _fn(self, name):
        return getattr(_capi.lib(), f"{self._prefix}_{name}")

    def __init__(self, device=-1):
        self._h = C.c_void_p()
        _capi.check(self._fn("create")(int(device), C.byref(self._h)))
        self._keep = None

    def close(self):
        if self._h:
            self._fn("destroy")(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        _capi.check(self._fn("set_option")(self._h, key.encode(), int(value)))

    def set_regions(self, desc_list, desc_bytes=None):
        """desc_list: sequence of (n_k, L) arrays (uint8 packed bits / float32 for L2fContext; n_k may be 0), one L for all."""
        arrs = [np.ascontiguousarray(d, dtype=self._dtype) for d in desc_list]
        if desc_bytes is None:
            lens = {a.shape[1] for a in arrs if a.ndim == 2 and a.shape[0]}
            if len(lens) > 1:
                raise ValueError("all binary descriptors must have one length")
            desc_bytes = lens.pop() if lens else 64
        n = len(arrs)
        ptrs = (C.c_void_p * max(n, 1))()
        cnt = (C.c_uint32 * max(n, 1))()
        for k, a in enumerate(arrs):
            ptrs[k] = a.ctypes.data if a.size else None
            cnt[k] = a.shape[0] if a.size else 0
        self._keep = arrs
        _capi.check(self._fn("set_regions")(self._h, ptrs, cnt, n, int(desc_bytes)))

    def run(self, pairs, dist_ratio):
        """pairs: (n_pairs, 2) uint32. Returns (stats, offsets[n_pairs+1] uint64, ij[(n_matches, 2)] uint32)."""
        pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        st = _capi.MatchStats()
        _capi.check(self._fn("run")(self._h, pairs.ctypes.data, pairs.shape[0], np.float32(dist_ratio), C.byref(st)))
        po = C.POINTER(C.c_uint64)()
        pij = C.POINTER(C.c_uint32)()
        _capi.check(self._fn("results")(self._h, C.byref(po), C.byref(pij)))
        n = pairs.shape[0]
        offsets = np.ctypeslib.as_array(po, shape=(n + 1,)).copy()
        total = int(offsets[-1])
        ij = np.ctypeslib.as_array(pij, shape=(total, 2)).copy() if total else np.zeros((0, 2), np.uint32)
        return st, offsets, ij


class L2fContext(HammingContext):
    """Float descriptors (AKAZE_Float_Regions: 64 floats), BRUTE_FORCE_L2 (thin wrapper over mvgx_l2f_*); run() takes the
    squared ratio like MatchContext.run()."""
    _prefix = "mvgx_l2f"
    _dtype = np.float32


class L2u8Context(HammingContext):
    """uint8 descriptors of length 64 / 128 / 144 (AKAZE_Liop_Regions), BRUTE_FORCE_L2 in exact integers (mvgx_l2u8_*);
    run() takes the squared ratio like MatchContext.run()."""
    _prefix = "mvgx_l2u8"
    _dtype = np.uint8


class CascadeContext(HammingContext):
    """CASCADE_HASHING_L2, matching stage (thin wrapper over mvgx_cascade_*): uint8 descriptors of 128 / 144 bytes or float ones of
    length 64 with the hash codes and bucket ids the caller's hashing stage produced (the openMVG adapter runs the reference's CascadeHasher for that); run() takes
    the squared ratio and returns the lists before the reference's de-duplication steps."""
    _prefix = "mvgx_cascade"
    _dtype = np.uint8

    def set_regions(self, desc_list, hash_list, bucket_list, n_groups=6, bits_per_bucket=10, dtype=np.uint8, dim=128):
        """dtype / dim: np.uint8 with 128 (SIFT_Regions) or 144 (AKAZE_Liop_Regions), np.float32 with 64 (AKAZE_Float_Regions) -
        mvgx_cascade_set_regions_typed; the hash codes have one bit per dimension ((dim + 7) // 8 bytes per descriptor)."""
        is_float = np.dtype(dtype) == np.float32
        hb = (dim + 7) // 8
        d = [np.ascontiguousarray(x, np.float32 if is_float else np.uint8).reshape(-1, dim) for x in desc_list]
        h = [np.ascontiguousarray(x, np.uint8).reshape(-1, hb) for x in hash_list]
        b = [np.ascontiguousarray(x, np.uint16).reshape(-1, n_groups) for x in bucket_list]
        n = len(d)
        dp = (C.c_void_p * max(n, 1))(); hp = (C.c_void_p * max(n, 1))(); bp = (C.c_void_p * max(n, 1))()
        cnt = (C.c_uint32 * max(n, 1))()
        for k in range(n):
            assert len(h[k]) == len(d[k]) == len(b[k])
            dp[k] = d[k].ctypes.data if len(d[k]) else None
            hp[k] = h[k].ctypes.data if len(d[k]) else None
            bp[k] = b[k].ctypes.data if len(d[k]) else None
            cnt[k] = len(d[k])
        self._keep = (d, h, b)
        _capi.check(self._fn("set_regions_typed")(self._h, 1 if is_float else 0, dp, hp, bp, cnt, n, dim, hb, n_groups, bits_per_bucket))

    def hash_regions(self, desc_list, zero_mean=None, n_groups=6, bits_per_bucket=10, random_seed=5489, fetch=False, dtype=np.uint8, dim=128):
        """The hashing stage on the device (mvgx_cascade_hash_regions_typed) in place of set_regions: descriptors only; dtype / dim as in
        set_regions (uint8 128 / 144, float32 64). zero_mean defaults to cascade_zero_mean(desc_list) (128-byte rows; the other shapes
        pass the caller's CascadeHasher::GetZeroMeanDescriptor result). fetch=True also returns (hash codes [(n, (dim + 7) // 8) uint8],
        bucket ids [(n, groups) uint16]) per image, the shapes of the reference's HashedDescription."""
        is_float = np.dtype(dtype) == np.float32
        hb = (dim + 7) // 8
        d = [np.ascontiguousarray(x, np.float32 if is_float else np.uint8).reshape(-1, dim) for x in desc_list]
        if zero_mean is None:
            if is_float or dim != 128:
                raise ValueError("hash_regions: zero_mean is required for other shapes than 128-byte uint8 rows")
            zero_mean = cascade_zero_mean(d)
        zm = np.ascontiguousarray(zero_mean, np.float32).reshape(dim)
        n = len(d)
        dp = (C.c_void_p * max(n, 1))(); hp = (C.c_void_p * max(n, 1))(); bp = (C.c_void_p * max(n, 1))()
        cnt = (C.c_uint32 * max(n, 1))()
        h = [np.zeros((len(x), hb), np.uint8) for x in d] if fetch else None
        b = [np.zeros((len(x), n_groups), np.uint16) for x in d] if fetch else None
        for k in range(n):
            dp[k] = d[k].ctypes.data if len(d[k]) else None
            cnt[k] = len(d[k])
            if fetch:
                hp[k] = h[k].ctypes.data if len(d[k]) else None
                bp[k] = b[k].ctypes.data if len(d[k]) else None
        self._keep = (d, zm)
        _capi.check(self._fn("hash_regions_typed")(self._h, 1 if is_float else 0, dp, cnt, n, dim, zm.ctypes.data, n_groups, bits_per_bucket, random_seed,
                                                   hp if fetch else None, bp if fetch else None))
        return (h, b) if fetch else None


def cascade_zero_mean(desc_list):
    """The zero-mean descriptor of the reference's hashing stage (Cascade_Hashing_Matcher_Regions.cpp:78-104: the mean over the images
    of the per-image mean, both CascadeHasher::GetZeroMeanDescriptor = cast<float>().colwise().mean()) with Eigen 3.4's operation
    order in single precision, as an AVX build evaluates it (tests/test_cascade.py pins it against the compiled reference):
      * per image (row-major uint8 map, cast to float: no packet access): each column is summed row after row, then divided by n;
      * over the images (column-major MatrixXf, 32-byte aligned): each column is a contiguous run reduced by Eigen's linear vectorised
        redux - two 8-float accumulators over the aligned part, a third packet if one is left, the horizontal sum
        ((a0+a4)+(a2+a6))+((a1+a5)+(a3+a7)), then the leading and trailing scalars - and divided by the number of images."""
    f32 = np.float32
    n_img = len(desc_list)
    per = np.zeros((n_img, 128), f32)
    for k, d in enumerate(desc_list):
        d = np.asarray(d, np.uint8).reshape(-1, 128)
        if len(d) == 0:
            continue
        acc = d[0].astype(f32)
        for r in range(1, len(d)):
            acc = acc + d[r].astype(f32)   # float32 + float32: one rounding per step, per column
        per[k] = acc / f32(len(d))
    if n_img == 0:
        return np.zeros(128, f32)
    out = np.zeros(128, f32)
    ps = 8
    for c in range(128):
        col = per[:, c]
        size = n_img
        start = min((-(c * n_img)) % ps, size)   # floats up to the next 32-byte boundary of the column-major storage
        aligned_size2 = ((size - start) // (2 * ps)) * (2 * ps)
        aligned_size = ((size - start) // ps) * ps
        end2, end = start + aligned_size2, start + aligned_size
        if aligned_size:
            p0 = col[start:start + ps].copy()
            if aligned_size > ps:
                p1 = col[start + ps:start + 2 * ps].copy()
                for i in range(start + 2 * ps, end2, 2 * ps):
                    p0 = p0 + col[i:i + ps]
                    p1 = p1 + col[i + ps:i + 2 * ps]
                p0 = p0 + p1
                if end > end2:
                    p0 = p0 + col[end2:end2 + ps]
            b = p0[:4] + p0[4:]
            res = f32(f32(b[0] + b[2]) + f32(b[1] + b[3]))
            for i in range(0, start):
                res = f32(res + col[i])
            for i in range(end, size):
                res = f32(res + col[i])
        else:
            res = col[0]
            for i in range(1, size):
                res = f32(res + col[i])
        out[c] = f32(res) / f32(n_img)
    return out


class Float_Regions(Regions):
    """Scalar_Regions<SIOPointFeature, float, L> stand-in (AKAZE_Float_Regions: L = 64): an (n, L) float32 array."""

    def __init__(self, descriptors):
        d = np.ascontiguousarray(descriptors, dtype=np.float32)
        if d.ndim != 2:
            raise ValueError("descriptors must be a 2-D array (n, L)")
        self._d = d

    def Type_id(self):
        return "f"  # typeid(float).name()


class Matcher_Regions:
    """Drop-in mirror of matching_image_collection::Matcher_Regions for BRUTE_FORCE_L2 on SIFT-like uint8 regions and on
    64-D float regions, and BRUTE_FORCE_HAMMING on binary regions (MI355X paths)."""

    def __init__(self, distRatio, eMatcherType, device=-1, variant=None):
        self.f_dist_ratio_ = np.float32(distRatio)
        self.eMatcherType_ = EMatcherType(eMatcherType)
        self._device = device
        self._variant = variant

    def Match(self, regions_provider, pairs, map_PutativeMatches, my_progress_bar=None):
        if self.eMatcherType_ == EMatcherType.BRUTE_FORCE_HAMMING:
            return self._match_hamming(regions_provider, pairs, map_PutativeMatches, my_progress_bar)
        if self.eMatcherType_ != EMatcherType.BRUTE_FORCE_L2:
            raise NotImplementedError(
                f"{self.eMatcherType_.name}: only BRUTE_FORCE_L2 / BRUTE_FORCE_HAMMING are accelerated; use openMVG's own "
                "matcher for the rest")
        pairs = sorted(set((int(a), int(b)) for a, b in pairs))  # Pair_Set is an ordered std::set
        if my_progress_bar is not None:
            my_progress_bar.Restart(len(pairs), "- Matching -")
        if not pairs:
            return
        ids = sorted({v for p in pairs for v in p})
        regs = {}
        for v in ids:
            r = regions_provider.get(v)
            if r is None:
                raise KeyError(f"Regions_Provider has no regions for view {v}")
            regs[v] = r
        if any(r.RegionCount() and r.Type_id() == "f" for r in regs.values()):
            return self._match_float(regs, ids, pairs, map_PutativeMatches, my_progress_bar)
        lens = {r.DescriptorLength() for r in regs.values() if r.RegionCount() and r.Type_id() == "h" and r.IsScalar()}
        if lens and lens <= {64, 144} and len(lens) == 1:   # e.g. AKAZE_Liop_Regions: the integer VALU path
            return self._match_u8_other(regs, ids, pairs, lens.pop(), map_PutativeMatches, my_progress_bar)
        # Matcher_Regions.cpp:85-90: pairs whose Type_id differ are skipped; regions_matcher.cpp:75-81: uchar only here
        for v, r in regs.items():
            if r.RegionCount() and (r.Type_id() != "h" or r.DescriptorLength() != 128 or not r.IsScalar()):
                raise NotImplementedError("device path handles Scalar_Regions<uint8, 128> (SIFT_Regions) only")
        local = {v: k for k, v in enumerate(ids)}
        descs = [regs[v].DescriptorRawData() if regs[v].RegionCount() else np.zeros((0, 128), np.uint8) for v in ids]
        ctx = MatchContext(self._device)
        try:
            if self._variant is not None:
                ctx.set_option("variant", self._variant)
            ctx.set_regions(descs)
            parr = np.array([(local[a], local[b]) for a, b in pairs], dtype=np.uint32).reshape(-1, 2)
            ratio_sq = np.float32(self.f_dist_ratio_ * self.f_dist_ratio_)  # Square() in float, numeric.h:56
            _, offsets, ij = ctx.run(parr, ratio_sq)
        finally:
            ctx.close()
        for k, p in enumerate(pairs):
            a, b = int(offsets[k]), int(offsets[k + 1])
            if b > a:  # only non-empty vectors are inserted (Matcher_Regions.cpp:99-102)
                map_PutativeMatches.insert(p, ij[a:b].copy())
            if my_progress_bar is not None:
                my_progress_bar += 1

    def _match_hamming(self, regions_provider, pairs, map_PutativeMatches, my_progress_bar=None):
        """regions_matcher.cpp:184-191: binary regions, Hamming<unsigned char>, ratio not squared."""
        pairs = sorted(set((int(a), int(b)) for a, b in pairs))
        if my_progress_bar is not None:
            my_progress_bar.Restart(len(pairs), "- Matching -")
        if not pairs:
            return
        ids = sorted({v for p in pairs for v in p})
        regs = {}
        for v in ids:
            r = regions_provider.get(v)
            if r is None:
                raise KeyError(f"Regions_Provider has no regions for view {v}")
            if r.RegionCount() and (not r.IsBinary() or r.Type_id() != "h"):
                # RegionMatcherFactory returns no matcher for scalar regions + BRUTE_FORCE_HAMMING (regions_matcher.cpp:60-64)
                raise NotImplementedError("BRUTE_FORCE_HAMMING needs binary uint8 regions")
            regs[v] = r
        local = {v: k for k, v in enumerate(ids)}
        L = next((regs[v].DescriptorLength() for v in ids if regs[v].RegionCount()), 64)
        descs = [regs[v].DescriptorRawData() if regs[v].RegionCount() else np.zeros((0, L), np.uint8) for v in ids]
        ctx = HammingContext(self._device)
        try:
            ctx.set_regions(descs, L)
            parr = np.array([(local[a], local[b]) for a, b in pairs], dtype=np.uint32).reshape(-1, 2)
            _, offsets, ij = ctx.run(parr, self.f_dist_ratio_)
        finally:
            ctx.close()
        for k, p in enumerate(pairs):
            a, b = int(offsets[k]), int(offsets[k + 1])
            if b > a:
                map_PutativeMatches.insert(p, ij[a:b].copy())
            if my_progress_bar is not None:
                my_progress_bar += 1


    def _match_float(self, regs, ids, pairs, map_PutativeMatches, my_progress_bar=None):
        """regions_matcher.cpp:119-124: float regions, L2<float>, squared metric."""
        for v in ids:
            r = regs[v]
            if r.RegionCount() and (r.Type_id() != "f" or not r.IsScalar() or r.DescriptorLength() != 64):
                raise NotImplementedError("device path handles Scalar_Regions<float, 64> (AKAZE_Float_Regions) only")
        local = {v: k for k, v in enumerate(ids)}
        descs = [regs[v].DescriptorRawData() if regs[v].RegionCount() else np.zeros((0, 64), np.float32) for v in ids]
        ctx = L2fContext(self._device)
        try:
            ctx.set_regions(descs, 64)
            parr = np.array([(local[a], local[b]) for a, b in pairs], dtype=np.uint32).reshape(-1, 2)
            _, offsets, ij = ctx.run(parr, np.float32(self.f_dist_ratio_ * self.f_dist_ratio_))
        finally:
            ctx.close()
        for k, p in enumerate(pairs):
            a, b = int(offsets[k]), int(offsets[k + 1])
            if b > a:
                map_PutativeMatches.insert(p, ij[a:b].copy())
            if my_progress_bar is not None:
                my_progress_bar += 1

    def _match_u8_other(self, regs, ids, pairs, dim, map_PutativeMatches, my_progress_bar=None):
        """regions_matcher.cpp:75-81 on Scalar_Regions<uint8, dim != 128> (AKAZE_Liop_Regions: 144)."""
        for v in ids:
            r = regs[v]
            if r.RegionCount() and (r.Type_id() != "h" or not r.IsScalar() or r.DescriptorLength() != dim):
                raise NotImplementedError("one uint8 descriptor length per provider")
        local = {v: k for k, v in enumerate(ids)}
        descs = [regs[v].DescriptorRawData() if regs[v].RegionCount() else np.zeros((0, dim), np.uint8) for v in ids]
        ctx = L2u8Context(self._device)
        try:
            ctx.set_regions(descs, dim)
            parr = np.array([(local[a], local[b]) for a, b in pairs], dtype=np.uint32).reshape(-1, 2)
            _, offsets, ij = ctx.run(parr, np.float32(self.f_dist_ratio_ * self.f_dist_ratio_))
        finally:
            ctx.close()
        for k, p in enumerate(pairs):
            a, b = int(offsets[k]), int(offsets[k + 1])
            if b > a:
                map_PutativeMatches.insert(p, ij[a:b].copy())
            if my_progress_bar is not None:
                my_progress_bar += 1
