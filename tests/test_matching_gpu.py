"""GPU parity tests: the HIP path (through the C ABI) must reproduce the reference bit-exactly.

Checkers: the committed golden fixtures (outputs of the reference's Matcher_Regions), the C oracle on seeded inputs,
and size-independent properties at the full 2000-descriptor size."""
import ctypes as C

import numpy as np
import pytest

from openmvg_amd import _capi, matching, synth
from tests import _golden, _oracle
from tests.test_oracle_matching import _adversarial_set

pytestmark = pytest.mark.gpu

# 0 naive check kernel; 1..3 exact top-2 MFMA kernel with the three window-staging forms;
# 41..43 = variant 4 (filter + verify, the default) with staging form 1..3
# 73 = staging form 3 with the three-workgroups-per-CU 16x16x64 filter (l2_filter16h_kernel: two query tiles per wave, half windows)
# 43 = the default since round 5: staging form 3 with the 16x16x64 filter (l2_filter16_kernel); 63 = staging form 3 with the 32x32x32 filter
VARIANTS = [0, 1, 2, 3, 41, 42, 43, 48, 56, 63, 73]   # 4x: variant 4 with staging form x; 48 / 56: form 3 with the earlier epilogue forms


def run_hip(imgs, pairs, ratio, variant, batch_pairs=None):
    ctx = matching.MatchContext(0)
    try:
        if variant in (48, 56):
            ctx.set_option("variant", 4)
            ctx.set_option("stage", 3)
            ctx.set_option("debug_filter", variant - 40)
        elif variant in (63, 73):
            ctx.set_option("variant", 4)
            ctx.set_option("stage", 3)
            ctx.set_option("filter_shape", 32 if variant == 63 else 17)
        elif variant >= 40:
            ctx.set_option("variant", 4)
            ctx.set_option("stage", variant - 40)
            ctx.set_option("filter_shape", 16)
        else:
            ctx.set_option("variant", variant)
        if batch_pairs:
            ctx.set_option("batch_pairs", batch_pairs)
        ctx.set_regions(imgs)
        r = np.float32(ratio)
        st, offsets, ij = ctx.run(pairs, r * r)
    finally:
        ctx.close()
    return st, offsets, ij


def assert_same(pairs, offsets, ij, ref):
    got = _oracle.offsets_to_dict(pairs, offsets, ij)
    assert set(got) == set(ref), (sorted(set(got) ^ set(ref))[:10])
    for k in ref:
        assert np.array_equal(got[k], ref[k]), (k, got[k][:5], ref[k][:5])


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("case", _golden.CASES)
def test_golden_fixtures(case, variant):
    imgs, pairs, ratio, ref = _golden.load_case(case)
    st, offsets, ij = run_hip(imgs, pairs, ratio, variant)
    assert_same(pairs, offsets, ij, ref)
    assert int(st.n_matches) == sum(len(v) for v in ref.values())


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("ratio", [0.8, 0.6, 1.0])
def test_adversarial_vs_oracle(variant, ratio):
    imgs = _adversarial_set()
    n = len(imgs)
    pairs = np.concatenate([matching.exhaustive_pairs_array(n), matching.exhaustive_pairs_array(n)[:, ::-1]])
    o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pairs, ratio)
    _, offsets, ij = run_hip(imgs, pairs, ratio, variant)
    assert np.array_equal(offsets, o_off)
    assert np.array_equal(ij, o_ij)


@pytest.mark.parametrize("variant", VARIANTS)
def test_ragged_sizes_full_range_bytes(variant):
    """Tile/window boundaries (31/32/33, 255/256/257, 511/513) on iid uniform bytes (largest norms)."""
    sizes = [0, 1, 2, 3, 31, 32, 33, 63, 255, 256, 257, 511, 513, 1000]
    imgs = synth.random_descriptors(len(sizes), sizes, seed=12)
    # plant near-duplicates so that matches exist across ragged images
    rng = np.random.default_rng(2)
    for k in range(4, len(sizes)):
        m = min(sizes[k], sizes[k - 1])
        imgs[k][:m] = np.clip(imgs[k - 1][:m].astype(np.int16) + rng.integers(-9, 10, (m, 128)), 0, 255).astype(np.uint8)
    n = len(imgs)
    pairs = np.concatenate([matching.exhaustive_pairs_array(n), matching.exhaustive_pairs_array(n)[:, ::-1]])
    o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pairs, 0.8)
    _, offsets, ij = run_hip(imgs, pairs, 0.8, variant)
    assert int(o_off[-1]) > 500
    assert np.array_equal(offsets, o_off)
    assert np.array_equal(ij, o_ij)


@pytest.mark.parametrize("variant", [1, 2, 3, 41, 43])
def test_extreme_values_do_not_overflow_packed_keys(variant):
    """All-0 vs all-255 rows give d = 8 323 200 (the maximum); mixed extremes exercise every key range."""
    rng = np.random.default_rng(7)
    a = np.zeros((300, 128), np.uint8); b = np.full((300, 128), 255, np.uint8)
    a[::3] = rng.integers(0, 2, (100, 128), dtype=np.uint8) * 255
    b[::5] = rng.integers(0, 2, (60, 128), dtype=np.uint8) * 255
    c = rng.integers(0, 256, (300, 128), dtype=np.uint8)
    imgs = [a, b, c]
    pairs = np.array([[0, 1], [1, 0], [0, 2], [2, 0], [1, 2], [2, 1]], np.uint32)
    o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pairs, 1.0)
    _, offsets, ij = run_hip(imgs, pairs, 1.0, variant)
    assert np.array_equal(offsets, o_off) and np.array_equal(ij, o_ij)


@pytest.mark.parametrize("variant", [1, 3, 41, 43])
def test_rootsift_like_2000_desc_sampled_vs_oracle_and_batching(variant):
    """Full-size images (2000 x 128): every pair of 12 images vs the oracle; tiny batches exercise the batch seams."""
    imgs = synth.image_descriptors(12, n_desc=2000, seed=5)
    pairs = matching.exhaustive_pairs_array(12)
    o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pairs, 0.8)
    st, offsets, ij = run_hip(imgs, pairs, 0.8, variant, batch_pairs=7)
    assert int(st.n_desc_pairs) == 66 * 2000 * 2000
    assert int(o_off[-1]) > 1000
    assert np.array_equal(offsets, o_off)
    assert np.array_equal(ij, o_ij)


@pytest.mark.parametrize("variant", [1, 3, 41, 43])
def test_properties_at_full_size(variant):
    """Size-independent properties at 2000 descriptors/image:
    (1) an image against a row-permuted copy of itself matches every distinct row to its preimage;
    (2) permuting the database rows permutes the reported i indices and nothing else;
    (3) the result does not depend on the batch size."""
    rng = np.random.default_rng(42)
    base = synth.image_descriptors(3, n_desc=2000, seed=77)
    A = base[0]
    _, inv = np.unique(A, axis=0, return_inverse=True)
    assert len(set(inv.tolist())) == len(A)  # all rows distinct in this fixture
    perm = rng.permutation(len(A))
    Ap = A[perm]                                # Ap[k] = A[perm[k]]
    imgs = [A, Ap, base[1], base[2]]
    pairs = np.array([[0, 1], [1, 0], [0, 2], [1, 2], [2, 3]], np.uint32)
    _, off, ij = run_hip(imgs, pairs, 0.8, variant)
    g = _oracle.offsets_to_dict(pairs, off, ij)
    # (1) database A, queries Ap: query j matches i = perm[j], all 2000 accepted (d0 = 0 < 0.64 d1)
    m01 = g[(0, 1)]
    assert len(m01) == 2000 and np.array_equal(m01[:, 1], np.arange(2000)) and np.array_equal(m01[:, 0], perm)
    # (2) same queries (image 2) against A and against Ap: j lists equal, i lists related by the permutation
    m02, m12 = g[(0, 2)], g[(1, 2)]
    assert np.array_equal(m02[:, 1], m12[:, 1])
    assert np.array_equal(m02[:, 0], perm[m12[:, 0]])
    # (3) batch seams
    _, off2, ij2 = run_hip(imgs, pairs, 0.8, variant, batch_pairs=2)
    assert np.array_equal(off, off2) and np.array_equal(ij, ij2)


def _force_norm_parity(d, parity):
    """Flip the LSB of byte 0 where needed so that every row's sum (a - 128)^2 has the given parity."""
    d = d.copy()
    odd = (((d.astype(np.int64) - 128) ** 2).sum(axis=1) & 1).astype(bool)
    d[odd != bool(parity), 0] ^= 1
    return d


@pytest.mark.parametrize("variant", [1, 41, 42, 43])
def test_norm_parity_skew_and_cell_collisions(variant):
    """The tile layout partitions rows by the parity of their squared norm: all-even / all-odd / lopsided images double
    the tile count of one half and leave the other half all pad slots. Near-duplicate rows placed in consecutive
    original positions fall into the same (P-class, window) cell and exercise the verify stage's runner-up search."""
    rng = np.random.default_rng(99)
    base = synth.image_descriptors(4, n_desc=700, seed=31)
    even = _force_norm_parity(base[0], 0)
    odd = _force_norm_parity(base[1], 1)
    lop = base[2].copy(); lop[:600] = _force_norm_parity(lop[:600], 0)
    near = base[3].copy()
    for k in range(0, 600, 2):                      # row k+1 = row k with a few +-1 / +-2 nudges: d(k, k+1) small
        near[k + 1] = near[k]
        idx = rng.integers(0, 128, 3)
        near[k + 1, idx] = np.clip(near[k + 1, idx].astype(np.int64) + rng.integers(-2, 3, 3), 0, 255)
    tiny = [base[0][:2], base[1][:3], base[2][:17], base[3][:33]]
    imgs = [even, odd, lop, near, base[0]] + tiny
    pairs = np.array([(i, j) for i in range(len(imgs)) for j in range(len(imgs)) if i != j], np.uint32)
    for ratio in (0.8, 1.0, 0.95):
        o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pairs, ratio)
        _, offsets, ij = run_hip(imgs, pairs, ratio, variant)
        assert np.array_equal(offsets, o_off) and np.array_equal(ij, o_ij), ratio


def test_oneshot_sink_entry_point_matches_reference_container_semantics():
    """mvgx_match_pairs_u8_l2 + sink: only non-empty pairs are reported, in input order, ascending j."""
    imgs, pairs, ratio, ref = _golden.load_case("adv08")
    arrs = [np.ascontiguousarray(i) for i in imgs]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data if len(a) else None for a in arrs])
    cnt = (C.c_uint32 * len(arrs))(*[len(a) for a in arrs])
    seen = []

    def sink(_u, I, J, pij, n):
        seen.append(((int(I), int(J)), np.ctypeslib.as_array(pij, shape=(int(n), 2)).copy()))

    cb = _capi.MATCH_SINK(sink)
    r = np.float32(ratio)
    _capi.check(_capi.lib().mvgx_match_pairs_u8_l2(ptrs, cnt, len(arrs), 128, pairs.ctypes.data, len(pairs), r * r, 0, cb, None))
    assert [k for k, _ in seen] == [tuple(map(int, p)) for p in pairs if tuple(map(int, p)) in ref]
    for k, v in seen:
        assert np.array_equal(v, ref[k])
        assert np.all(np.diff(v[:, 1].astype(np.int64)) > 0)


def test_matcher_regions_mirror_drop_in():
    """The host-side mirror keeps the reference call shape: Matcher_Regions(ratio, BRUTE_FORCE_L2).Match(provider, pairs, map)."""
    imgs, pairs, ratio, ref = _golden.load_case("sift08")
    provider = matching.Regions_Provider({k: matching.Regions(d) for k, d in enumerate(imgs)})
    out = matching.PairWiseMatches()
    matching.Matcher_Regions(ratio, matching.EMatcherType.BRUTE_FORCE_L2).Match(provider, [tuple(p) for p in pairs], out)
    assert set(out) == set(ref)
    for k in ref:
        assert np.array_equal(out[k], ref[k])


def test_error_behaviour():
    ctx = matching.MatchContext(0)
    try:
        with pytest.raises(_capi.MvgxError) as e:
            ctx.set_regions([np.zeros((4, 64), np.uint8)])
        assert e.value.code == _capi.MVGX_ERR_UNSUPPORTED
        ctx.set_regions(synth.random_descriptors(2, 40, seed=1))
        with pytest.raises(_capi.MvgxError) as e:
            ctx.run(np.array([[0, 1]], np.uint32), 1.5)  # ratio^2 > 1: tie order is libstdc++-specific in the reference
        assert e.value.code == _capi.MVGX_ERR_UNSUPPORTED
        with pytest.raises(_capi.MvgxError) as e:
            ctx.run(np.array([[0, 2]], np.uint32), 0.64)
        assert e.value.code == _capi.MVGX_ERR_ARG
        st, off, ij = ctx.run(np.zeros((0, 2), np.uint32), 0.64)
        assert int(st.n_pairs) == 0 and len(ij) == 0
    finally:
        ctx.close()
    with pytest.raises(NotImplementedError):
        matching.Matcher_Regions(0.8, matching.EMatcherType.ANN_L2).Match(matching.Regions_Provider(), [(0, 1)], {})


@pytest.mark.parametrize("pinned", [1, 0])
def test_results_of_a_run_survive_the_next_run(pinned):
    """include/mvgx.h, "double_buffer_results": the context alternates between two result buffers, so the lists of run k stay valid while run k + 1
    executes (the adapter fills the match container from them on another thread); both pinned and plain result memory"""
    import ctypes as C
    from openmvg_amd import _capi
    imgs = synth.image_descriptors(5, n_desc=400, seed=77)
    pa = matching.exhaustive_pairs_array(5)
    pb = np.ascontiguousarray(pa[::-1, ::-1])
    ctx = matching.MatchContext(0)
    try:
        ctx.set_option("pinned_results", pinned)
        ctx.set_option("double_buffer_results", 1)
        ctx.set_regions(imgs)
        L = _capi.lib()

        def run(pairs):
            st = _capi.MatchStats()
            _capi.check(L.mvgx_match_run(ctx._h, pairs.ctypes.data, len(pairs), np.float32(0.64), C.byref(st)))
            po, pij = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint32)()
            _capi.check(L.mvgx_match_results(ctx._h, C.byref(po), C.byref(pij)))
            off = np.ctypeslib.as_array(po, shape=(len(pairs) + 1,))
            return off, np.ctypeslib.as_array(pij, shape=(int(off[-1]), 2))   # views, not copies

        off_a, ij_a = run(pa)
        keep_off, keep_ij = off_a.copy(), ij_a.copy()
        off_b, ij_b = run(pb)
        assert np.array_equal(off_a, keep_off) and np.array_equal(ij_a, keep_ij)      # run k still readable after run k + 1
        o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pa, 0.8)
        assert np.array_equal(keep_off, o_off) and np.array_equal(keep_ij, o_ij)
        o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pb, 0.8)
        assert np.array_equal(off_b, o_off) and np.array_equal(ij_b, o_ij)
    finally:
        ctx.close()


def test_match_directory_file_level_pipeline(tmp_path):
    """io.match_directory: <stem>.desc files in, matches.putative.txt out, through the Matcher_Regions
    mirror on the device; the file read back equals the oracle's lists"""
    from openmvg_amd import io
    sizes = [400, 0, 300, 1, 257, 2]
    imgs = synth.random_descriptors(len(sizes), sizes, seed=5)
    rng = np.random.default_rng(1)
    for k in (2, 4):      # near-duplicates of image 0 so that matches exist
        m = min(sizes[k], sizes[0])
        imgs[k][:m] = np.clip(imgs[0][:m].astype(np.int16) + rng.integers(-6, 7, (m, 128)), 0, 255).astype(np.uint8)
    stems = [f"view_{k}" for k in range(len(sizes))]
    for stem, d in zip(stems, imgs):
        io.save_desc_bin(str(tmp_path / (stem + ".desc")), d)
    got = io.match_directory(str(tmp_path), stems, ratio=0.8, kind="sift", device=0)
    back = io.load_matches_txt(str(tmp_path / "matches.putative.txt"))
    pairs = matching.exhaustive_pairs_array(len(sizes))
    off, ij = _oracle.port_matcher_regions_match(imgs, pairs, 0.8)
    want = _oracle.offsets_to_dict(pairs, off, ij)
    assert len(want) >= 2 and back.keys() == want.keys() == got.keys()
    assert all(np.array_equal(back[k], want[k]) for k in want)


# ---- parity at the benchmarked shape ("bit-exact comparison ... all pairs for n <= 64 images") --------------
@pytest.mark.skipif(not _oracle.have_ref_match(), reason="oracle/_ref (the compiled reference) not built")
def test_c2_shape_64_images_all_pairs_vs_reference():
    """64 images x 2000 descriptors of the bench.py data set (same generator and seed as BASELINE configs[1]), all 2016 pairs,
    default kernel and batch pipeline, against the reference's own Matcher_Regions::Match compiled in place (oracle/_ref)."""
    imgs = synth.image_descriptors(64, n_desc=2000, seed=0xC0FFEE00)
    pairs = matching.exhaustive_pairs_array(64)
    ref = _oracle.ref_matcher_regions_match(imgs, pairs, 0.8)
    # 500 pairs per batch: five batches, so the two-slot pipeline and the batch seams are part of what is compared
    st, offsets, ij = run_hip(imgs, pairs, 0.8, 43, batch_pairs=500)
    assert_same(pairs, offsets, ij, ref)
    assert int(st.n_desc_pairs) == 2016 * 2000 * 2000
    assert int(st.n_matches) == sum(len(v) for v in ref.values()) > 0


@pytest.mark.parametrize("n", [9000, 30000, 70001])
def test_large_images_vs_oracle(n):
    """Real SIFT gives 10-40 k features per image: query strides beyond one batch slot's default scratch, the batch size
    shrinks (mvgx_match_run's scratch cap); lists must stay bit-identical to the restatement."""
    imgs = synth.image_descriptors(3, n_desc=n, seed=11)
    imgs[2] = imgs[2][: n // 3 + 1]
    pairs = np.array([[0, 1], [1, 0], [0, 2], [2, 1]], np.uint32)
    o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pairs, 0.8)
    _, offsets, ij = run_hip(imgs, pairs, 0.8, 43)
    assert np.array_equal(offsets, o_off)
    assert np.array_equal(ij, o_ij)
    assert int(o_off[-1]) > 0


# ---- streaming hand-over and several device contexts in one process ------------------
@pytest.mark.parametrize("devices", [None, [0, 0], [0, 0, 0]])
def test_streamed_and_multi_device_runs_equal_the_single_run(devices):
    """mvgx_match_run_stream + mvgx_match_create_multi: 48 images x 2000 descriptors, 1128 pairs in batches of 100, two /
    three contexts on this one GPU (one host thread each, batches shared out dynamically): per-batch lists and the collected
    run equal the single-context run bit for bit."""
    imgs = synth.image_descriptors(48, n_desc=2000, seed=0xC0FFEE00)
    imgs[7] = imgs[7][:0]
    imgs[11] = imgs[11][:1]
    pairs = matching.exhaustive_pairs_array(48)
    _, want_off, want_ij = run_hip(imgs, pairs, 0.8, 43)
    ctx = matching.MatchContext(0) if devices is None else matching.MatchContext(devices=devices)
    try:
        ctx.set_option("batch_pairs", 100)
        ctx.set_regions(imgs)
        st, off, ij, firsts = ctx.run_collect_stream(pairs, np.float32(0.8) * np.float32(0.8))
        assert firsts == list(range(0, len(pairs), 100))
        assert np.array_equal(off, want_off) and np.array_equal(ij, want_ij)
        assert int(st.n_matches) == len(want_ij) and int(st.n_desc_pairs) == sum(len(imgs[a]) * len(imgs[b]) for a, b in pairs if len(imgs[a]) >= 2)
        st2, off2, ij2 = ctx.run(pairs, np.float32(0.8) * np.float32(0.8))
        assert np.array_equal(off2, want_off) and np.array_equal(ij2, want_ij)
    finally:
        ctx.close()


def test_streaming_keeps_host_memory_flat():
    """A 400-image run (79 800 pairs, ~0.16 GB of match lists) streamed in batches of 4096 pairs: the process's resident set
    grows by the two pinned batch buffers, not by the run's lists (what a 10 000-image run needs: DESIGN 3.2)."""
    import psutil
    imgs = synth.image_descriptors(400, n_desc=2000, seed=0xC0FFEE00)
    pairs = matching.exhaustive_pairs_array(400)
    ctx = matching.MatchContext(0)
    try:
        ctx.set_option("batch_pairs", 4096)
        ctx.set_regions(imgs)
        r2 = np.float32(0.8) * np.float32(0.8)
        tot = [0, 0]

        def on_batch(p0, off, lists):
            tot[0] += len(lists)
            tot[1] = max(tot[1], lists.nbytes)
        ctx.run_stream(pairs[:8192], r2, on_batch)   # first batches: buffers allocated
        rss0 = psutil.Process().memory_info().rss
        tot[0] = 0
        st = ctx.run_stream(pairs, r2, on_batch)
        rss1 = psutil.Process().memory_info().rss
        assert tot[0] == int(st.n_matches) > 10_000_000
        assert rss1 - rss0 < 4 * tot[1] + (32 << 20), (rss0, rss1, tot)
        assert 8 * tot[0] > 20 * (rss1 - rss0)    # the whole run's lists are >20x what the process grew by
    finally:
        ctx.close()


def test_profile_level_2_counts_the_filter_candidates():
    """ "profile" 2 adds the statistics pass: candidates handed to the verify stage (reported in kernel_vgprs) are at least the
    accepted matches and at most the queries; level 1 (what bench.py times) does not run it"""
    imgs = synth.image_descriptors(3, n_desc=200, seed=4)
    p = matching.exhaustive_pairs_array(3)
    pairs = np.concatenate([p, p[:, ::-1]])
    for level, counted in ((1, False), (2, True)):
        ctx = matching.MatchContext(0)
        try:
            ctx.set_option("profile", level)
            ctx.set_regions(imgs)
            st, off, ij = ctx.run(pairs, np.float32(0.8) * np.float32(0.8))
        finally:
            ctx.close()
        if counted:
            assert int(st.n_matches) <= int(st.kernel_vgprs) <= 200 * len(pairs)
        else:
            assert int(st.kernel_vgprs) == 0
