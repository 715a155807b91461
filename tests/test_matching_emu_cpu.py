"""CPU tests of the uint8 matching path's DEVICE code under the HIP execution-model emulation (tests/native/hipemu,
tests/_emu.py: second library, built from openmvg_amd/csrc/mvgx_match.hip with the LDS-DMA staging helpers replaced by per-lane
copies). The emulation implements v_mfma_i32_32x32x32_i8 with the gfx950 fragment maps, v_dot4_i32_i8, the DPP row
permutations, ballots and shuffles; tile layout, parity partition, the max3 filter epilogue, the verify stage, the ordered
compaction and the two-slot batch pipeline are the product's code. What this cannot check is gfx950 code generation, the
hand-placed waitcnt of the asynchronous staging and timing - the `-m gpu` tests run the real thing on the MI355X."""
import ctypes as C

import numpy as np
import pytest

from openmvg_amd import _capi, matching, synth
from tests import _emu, _golden, _oracle
from tests.test_matching_gpu import VARIANTS, assert_same, run_hip
from tests.test_oracle_matching import _adversarial_set


def _both_directions(n):
    p = matching.exhaustive_pairs_array(n)
    return np.concatenate([p, p[:, ::-1]])


@pytest.mark.parametrize("variant", VARIANTS)
def test_ragged_sizes_all_kernel_variants(variant):
    """tile / window boundaries (31 / 33 rows, 255 / 257: one LDS window more, 600: three windows per parity half) on
    full-range bytes, every kernel variant and staging form, against the C restatement"""
    sizes = [0, 1, 2, 3, 31, 33, 255, 257, 600]
    imgs = synth.random_descriptors(len(sizes), sizes, seed=12)
    rng = np.random.default_rng(2)
    for k in range(4, len(sizes)):
        m = min(sizes[k], sizes[k - 1])
        imgs[k][:m] = np.clip(imgs[k - 1][:m].astype(np.int16) + rng.integers(-9, 10, (m, 128)), 0, 255).astype(np.uint8)
    pairs = _both_directions(len(sizes))
    o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pairs, 0.8)
    with _emu.emulated():
        _, off, ij = run_hip(imgs, pairs, 0.8, variant)
    assert int(o_off[-1]) > 1000 and np.array_equal(off, o_off) and np.array_equal(ij, o_ij)


@pytest.mark.parametrize("case", _golden.CASES)
def test_golden_fixtures_of_the_reference(case):
    imgs, pairs, ratio, ref = _golden.load_case(case)
    with _emu.emulated():
        st, off, ij = run_hip(imgs, pairs, ratio, 41)
    assert_same(pairs, off, ij, ref)
    assert int(st.n_matches) == sum(len(v) for v in ref.values())


@pytest.mark.parametrize("ratio", [0.8, 1.0])
def test_adversarial_set(ratio):
    """all-zero rows, exact duplicates, duplicate nearest neighbours, nI in {0, 1, 2}, nJ = 0"""
    imgs = _adversarial_set()
    pairs = _both_directions(len(imgs))
    o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pairs, ratio)
    with _emu.emulated():
        for variant in (43, 48, 1):
            _, off, ij = run_hip(imgs, pairs, ratio, variant)
            assert np.array_equal(off, o_off) and np.array_equal(ij, o_ij), variant


def test_extreme_values_and_parity_skew():
    """d = 8 323 200 (all-0 vs all-255), images whose rows all have even / odd squared norm (one parity half stays empty),
    near-duplicates inside one 16-row cell (the verify stage's runner-up search)"""
    rng = np.random.default_rng(7)
    a = np.zeros((150, 128), np.uint8); b = np.full((150, 128), 255, np.uint8)
    a[::3] = rng.integers(0, 2, (50, 128), dtype=np.uint8) * 255
    b[::5] = rng.integers(0, 2, (30, 128), dtype=np.uint8) * 255
    base = synth.image_descriptors(2, n_desc=300, seed=31)
    from tests.test_matching_gpu import _force_norm_parity
    even = _force_norm_parity(base[0], 0)
    near = base[1].copy()
    for k in range(0, 280, 2):
        near[k + 1] = near[k]
        idx = rng.integers(0, 128, 3)
        near[k + 1, idx] = np.clip(near[k + 1, idx].astype(np.int64) + rng.integers(-2, 3, 3), 0, 255)
    imgs = [a, b, even, near, base[0]]
    pairs = np.array([(i, j) for i in range(5) for j in range(5) if i != j], np.uint32)
    with _emu.emulated():
        for ratio in (1.0, 0.8):
            o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pairs, ratio)
            for variant in (41, 48):
                _, off, ij = run_hip(imgs, pairs, ratio, variant)
                assert np.array_equal(off, o_off) and np.array_equal(ij, o_ij), (ratio, variant)


def test_batch_pipeline_options_and_result_buffers():
    """tiny batches through the two-slot pipeline, serial mode, plain-memory and double-buffered results: the lists are
    independent of all of them; run k stays readable after run k + 1 with "double_buffer_results" """
    imgs = synth.image_descriptors(5, n_desc=120, seed=77)
    pa = matching.exhaustive_pairs_array(5)
    pb = np.ascontiguousarray(pa[::-1, ::-1])
    oa = _oracle.port_matcher_regions_match(imgs, pa, 0.8)
    ob = _oracle.port_matcher_regions_match(imgs, pb, 0.8)
    with _emu.emulated():
        for opts in ({"batch_pairs": 3}, {"batch_pairs": 2, "overlap": 0}, {"batch_pairs": 4, "pinned_results": 0}):
            ctx = matching.MatchContext(0)
            for k, v in opts.items():
                ctx.set_option(k, v)
            ctx.set_regions(imgs)
            _, off, ij = ctx.run(pa, np.float32(0.64))
            ctx.close()
            assert np.array_equal(off, oa[0]) and np.array_equal(ij, oa[1]), opts
        ctx = matching.MatchContext(0)
        ctx.set_option("double_buffer_results", 1); ctx.set_option("batch_pairs", 3)
        ctx.set_regions(imgs)
        L = _capi.lib()

        def run(pairs):
            st = _capi.MatchStats()
            _capi.check(L.mvgx_match_run(ctx._h, pairs.ctypes.data, len(pairs), np.float32(0.64), C.byref(st)))
            po, pij = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint32)()
            _capi.check(L.mvgx_match_results(ctx._h, C.byref(po), C.byref(pij)))
            off = np.ctypeslib.as_array(po, shape=(len(pairs) + 1,))
            return off, np.ctypeslib.as_array(pij, shape=(int(off[-1]), 2))

        off_a, ij_a = run(pa)
        off_b, ij_b = run(pb)
        assert np.array_equal(off_a, oa[0]) and np.array_equal(ij_a, oa[1])      # still valid after the next run
        assert np.array_equal(off_b, ob[0]) and np.array_equal(ij_b, ob[1])
        ctx.close()


def test_oneshot_sink_and_mirror():
    """mvgx_match_pairs_u8_l2 (serial callback, ascending input order) and the Matcher_Regions mirror"""
    imgs = synth.image_descriptors(4, n_desc=90, seed=3)
    imgs[2] = imgs[2][:0]
    pairs = matching.exhaustive_pairs_array(4)
    o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pairs, 0.8)
    want = _oracle.offsets_to_dict(pairs, o_off, o_ij)
    with _emu.emulated():
        prov = matching.Regions_Provider({k: matching.Regions(d) for k, d in enumerate(imgs)})
        out = matching.PairWiseMatches()
        matching.Matcher_Regions(0.8, matching.EMatcherType.BRUTE_FORCE_L2).Match(prov, [tuple(p) for p in pairs], out)
    assert dict(out).keys() == want.keys() and all(np.array_equal(out[k], want[k]) for k in want)


@pytest.mark.parametrize("world", [2, 3])
def test_pair_shards_reproduce_the_single_rank_run(world):
    """the pair list is cut into per-rank shards (sharding.shard_pairs, balanced by descriptor pairs), every
    rank holds all descriptors and runs its shard with no collective; the shards' lists concatenated in rank order are the
    single-rank lists (emulated device code, one context per rank)"""
    from openmvg_amd import sharding
    sizes = [90, 40, 0, 130, 75, 20]
    imgs = synth.image_descriptors(len(sizes), n_desc=max(sizes), seed=19)
    imgs = [d[:s] for d, s in zip(imgs, sizes)]
    all_pairs = matching.exhaustive_pairs_array(len(sizes))
    o_off, o_ij = _oracle.port_matcher_regions_match(imgs, all_pairs, 0.8)
    got_pairs, got_ij, counts = [], [], []
    with _emu.emulated():
        for rank in range(world):
            mine = np.ascontiguousarray(sharding.shard_pairs(all_pairs, sizes, rank, world))
            _, off, ij = run_hip(imgs, mine, 0.8, 41)
            got_pairs.append(mine); got_ij.append(ij); counts.append(np.diff(off))
    assert np.array_equal(np.concatenate(got_pairs), all_pairs)          # contiguous shards, in order, covering every pair once
    assert np.array_equal(np.concatenate(counts), np.diff(o_off))
    assert np.array_equal(np.concatenate(got_ij), o_ij)


@pytest.mark.parametrize("devices", [None, [0, 0], [0, 0, 0]])
def test_streamed_and_multi_device_runs_equal_the_single_run(devices):
    """mvgx_match_run_stream (lists handed over batch by batch on the calling thread, host memory O(batch)) and multi-device
    contexts (mvgx_match_create_multi: one host thread per device, batches shared out dynamically, no collective):
    every batch carries the lists of the single-device run; several emulated contexts on 'device 0'"""
    sizes = [90, 40, 0, 130, 75, 20, 1, 64]
    imgs = synth.image_descriptors(len(sizes), n_desc=max(sizes), seed=23)
    imgs = [d[:s] for d, s in zip(imgs, sizes)]
    pairs = matching.exhaustive_pairs_array(len(sizes))
    o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pairs, 0.8)
    with _emu.emulated():
        for opts in ({"batch_pairs": 3}, {"batch_pairs": 5, "overlap": 0}, {"batch_pairs": 64}):
            ctx = matching.MatchContext(0) if devices is None else matching.MatchContext(devices=devices)
            for k, v in opts.items():
                ctx.set_option(k, v)
            ctx.set_regions(imgs)
            st, off, ij, firsts = ctx.run_collect_stream(pairs, np.float32(0.64))
            assert np.array_equal(off, o_off) and np.array_equal(ij, o_ij), (devices, opts)
            assert st.n_matches == len(o_ij)
            b = opts["batch_pairs"]
            assert firsts == list(range(0, len(pairs), b))   # every batch was handed over exactly once
            # the collecting entry point on the same context
            _, off2, ij2 = ctx.run(pairs, np.float32(0.64))
            assert np.array_equal(off2, o_off) and np.array_equal(ij2, o_ij), (devices, opts)
            # a sink that asks to stop is not entered again
            seen = []
            ctx.run_stream(pairs, np.float32(0.64), lambda p0, off_, ij_: seen.append(p0) or True)
            assert len(seen) == 1
            ctx.close()


def test_devices_from_the_environment(monkeypatch):
    """mvgx_match_create(-1) consults MVGX_DEVICES ("all", a list of ordinals); a bad list is an argument error"""
    imgs = synth.image_descriptors(3, n_desc=50, seed=5)
    pairs = matching.exhaustive_pairs_array(3)
    o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pairs, 0.8)
    with _emu.emulated():
        for env in ("0,0", "all", "0"):
            monkeypatch.setenv("MVGX_DEVICES", env)
            ctx = matching.MatchContext(-1)
            ctx.set_option("batch_pairs", 2)
            ctx.set_regions(imgs)
            _, off, ij = ctx.run(pairs, np.float32(0.64))
            ctx.close()
            assert np.array_equal(off, o_off) and np.array_equal(ij, o_ij), env
        monkeypatch.setenv("MVGX_DEVICES", "0,99")
        with pytest.raises(_capi.MvgxError):
            matching.MatchContext(-1)


@pytest.mark.parametrize("devices", [None, [0, 0]])
def test_stream_hold_keeps_a_batch_valid_for_two_further_sink_calls(devices):
    """option "stream_hold": the arrays handed to the sink at call k are still intact when calls k + 1 and k + 2 are made (second
    set of host buffers per slot) - what the openMVG adapter relies on to build its containers off the calling thread"""
    sizes = [90, 40, 130, 75, 20, 64, 33, 51]
    imgs = synth.image_descriptors(len(sizes), n_desc=max(sizes), seed=29)
    imgs = [d[:s] for d, s in zip(imgs, sizes)]
    pairs = matching.exhaustive_pairs_array(len(sizes))
    o_off, o_ij = _oracle.port_matcher_regions_match(imgs, pairs, 0.8)
    with _emu.emulated():
        ctx = matching.MatchContext(0) if devices is None else matching.MatchContext(devices=devices)
        ctx.set_option("batch_pairs", 3)
        ctx.set_option("stream_hold", 1)
        ctx.set_option("pinned_stream", 0)
        ctx.set_regions(imgs)
        held = []     # (first pair, live views, copies taken at hand-over)
        checked = [0]

        def on_batch(p0, off, lists):
            for q0, (voff, vij), (coff, cij) in held[-2:]:      # the two previous hand-overs
                assert np.array_equal(voff, coff) and np.array_equal(vij, cij), (q0, p0)
                checked[0] += 1
            held.append((p0, (off, lists), (off.copy(), lists.copy())))

        ctx.run_stream(pairs, np.float32(0.64), on_batch)
        for q0, (voff, vij), (coff, cij) in held[-2:]:          # ... and the last ones after the run has returned
            assert np.array_equal(voff, coff) and np.array_equal(vij, cij)
        ctx.close()
    assert checked[0] >= 2 * (len(held) - 2)
    got = np.concatenate([c[2][1] for c in sorted(held, key=lambda h: h[0])])
    assert np.array_equal(got, o_ij)

