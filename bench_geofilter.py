#!/usr/bin/env python
"""Side record of bench.py: the geometric filter (GeometricFilter_FMatrix_AC: 4 px, 2048 iterations - the
settings of main_GeometricFilter) on synthetic two-view correspondences, one MI355X, with the reference's own kernel + ACRANSAC
(oracle/_ref/libref_geofilter.so, OpenMP over the pairs like ImageCollectionGeometricFilter) timed beside it on a bounded sample
and the inlier sets of that sample compared pair by pair."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RESIDENT_WAVES = 256 * 4 * 3      # 256 CUs x 4 SIMDs x 3 waves (the kernel's 168 VGPRs; LDS would allow 6)
SHADER_CLOCK_HZ = 2.4e9           # nominal (MI355X_MICROARCH.md)
# pairs on which two builds of the reference (-O3 and -O3 -mavx2 -mfma) end with different inlier sets, bench sets (tools/geofilter_ref_vs_ref.py)
REF_VS_REF_FILE = "profiles/round4_geofilter_reference_vs_reference.json"


def _ref_vs_ref():
    try:
        with open(os.path.join(ROOT, REF_VS_REF_FILE)) as f:
            r = json.load(f)
        return {m: f"{r[m]['pairs_differing']} of {r[m]['pairs']} ({REF_VS_REF_FILE})" for m in ("f", "h", "e") if m in r}
    except (OSError, KeyError, ValueError):
        return {}


REF_VS_REF = _ref_vs_ref()
PMC_PROFILE = "profiles/round5_geofilter_pmc_call_r5_58.json"   # SQ counter passes of tools/geofilter_run.py (tools/gpu.sh geopmc) on the kernels as they are (E and H with samples ahead)


def valu_cycles_per_iteration(model):
    """SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / iterations of the committed counter pass: the kernel's own VALU issue time per iteration"""
    try:
        with open(os.path.join(ROOT, PMC_PROFILE)) as f:
            r = json.load(f)[model]
        return r["raw"]["b"]["SQ_ACTIVE_INST_VALU"] * 4.0 / r["iterations"]
    except (OSError, KeyError, ValueError, TypeError):
        return None


def _at_least_a_second(ref_fn, sub):
    """the reference on `sub`, repeated until a second of CPU time has been measured (a 0.01 s sample is noise); returns the first
    result with its "seconds" replaced by the mean over the repetitions, plus the number of "repetitions" made"""
    ref = ref_fn(sub)
    total, reps = ref["seconds"], 1
    while total < 1.0 and reps < 2000:
        total += ref_fn(sub)["seconds"]
        reps += 1
    ref["seconds"] = total / reps
    ref["repetitions"] = reps
    return ref


def geofilter_bench_record(device=0, n_pairs=100000, n=250, steps=2, cpu=True, cpu_pairs=6000, model="f"):
    """model "f": GeometricFilter_FMatrix_AC; "h": GeometricFilter_HMatrix_AC (H_ACRobust.hpp) on pairs related by homographies;
    "e": GeometricFilter_EMatrix_AC (E_ACRobust.hpp) on the calibrated version of the "f" set"""
    from openmvg_amd import geofilter, synth
    K = bear = None
    if model == "e":
        tv = synth.two_view_matches_bulk(n_pairs, n=n, seed=0x6E0F)
        K = synth.two_view_calibration(tv)
        fun = geofilter.GeometricFilter_EMatrix_AC(4.0, 2048)
        # one calibration for the whole set: the bearing vectors of all correspondences at once (what the cameras' operator() returns)
        bear = (geofilter.pinhole_bearings(K[0, 0], tv["xI"]), geofilter.pinhole_bearings(K[0, 1], tv["xJ"]))
    elif model == "h":
        tv = synth.two_view_homography_matches(n_pairs, seed=0x6E0F, n_min=n, n_max=n, tiny_frac=0.0)
        fun = geofilter.GeometricFilter_HMatrix_AC(4.0, 2048)
    else:
        tv = synth.two_view_matches_bulk(n_pairs, n=n, seed=0x6E0F)
        fun = geofilter.GeometricFilter_FMatrix_AC(4.0, 2048)
    def run(m):   # the first m pairs of the set
        if model == "e":
            return geofilter.filter_pairs_e(tv["xI"][:n * m], tv["xJ"][:n * m], tv["start"][:m + 1], tv["wh"][:m], K[:m], fun, device,
                                            bearings=(bear[0][:n * m], bear[1][:n * m]))
        return geofilter.filter_pairs(tv["xI"][:n * m], tv["xJ"][:n * m], tv["start"][:m + 1], tv["wh"][:m], fun, device)

    run(min(512, n_pairs))   # warm-up
    run(n_pairs)             # ... and one untimed call at full size: the library's device / page-locked slab caches are sized by the first such call
    # the dependent chain of one iteration without contention: 256 pairs = 64 workgroups of four waves, one wave per SIMD on 64 CUs
    _, _, st1 = run(min(256, n_pairs))
    chain_clocks = st1.wave_clocks / max(int(st1.n_iterations), 1)
    kernel_ms = total_ms = 0.0
    iters = models = clocks = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        mask, res, st = run(n_pairs)
        kernel_ms += st.kernel_ms; total_ms += st.total_ms
        iters += int(st.n_iterations); models += int(st.n_models); clocks += int(st.wave_clocks)
    dt = time.perf_counter() - t0
    ach = iters / (kernel_ms * 1e-3)
    resident = RESIDENT_WAVES * 2 // 3 if model == "e" else RESIDENT_WAVES   # the essential instantiation holds two waves per SIMD (256 VGPRs)
    peak = resident * SHADER_CLOCK_HZ / chain_clocks
    vci = valu_cycles_per_iteration(model)
    rec = {"metric": f"image pairs/s (a-contrario {'homography' if model == 'h' else 'essential-matrix' if model == 'e' else 'fundamental-matrix'} filter of putative matches)", "value": n_pairs * steps / (total_ms * 1e-3),
           "unit": "image pairs/s (whole call: host preparation, transfers, kernels)", "dtype": "f64",
           "image_pairs_per_s_kernel_time": n_pairs * steps / (kernel_ms * 1e-3),
           "roofline": {"bound": "latency", "achieved": ach, "peak": peak, "unit": "a-contrario iterations/s", "frac": ach / peak, "traffic": None,
                        "kernel": f"geofilter_f_acransac_kernel<4, false, {'kModelH' if model == 'h' else 'kModelE' if model == 'e' else 'kModelF'}>",
                        "iterations_per_pass": iters / steps, "models_per_iteration": models / max(iters, 1),
                        "clocks_per_iteration_and_wave": clocks / max(iters, 1), "clocks_per_iteration_one_wave_per_simd": chain_clocks,
                        "resident_waves": resident, "clock_hz_nominal": SHADER_CLOCK_HZ,
                        "valu_cycles_per_iteration_pmc": vci,
                        "frac_of_valu_issue_floor": (ach * vci / (1024 * SHADER_CLOCK_HZ)) if vci else None,
                        "note": "one wave runs one pair's sequential program: the floor of an iteration is its dependent chain (sample -> minimal solver -> "
                                "residuals / histogram -> NFA), measured in this run with one wave per SIMD (s_memtime, first to last instruction of every "
                                "wave / iterations); peak = resident waves (3 per SIMD: 168 VGPRs) x clock / that chain. frac_of_valu_issue_floor = the same "
                                f"rate against 1024 SIMDs x clock / the VALU issue cycles of an iteration, from the SQ counter pass {PMC_PROFILE}"},
           "config": {"workload": f"{n_pairs} image pairs x {n} putative matches (25 % of the pairs without geometry, the others 30-90 % inliers, "
                                  f"0.4 px noise), precision 4 px, 2048 iterations", "pairs_accepted": int(st.n_pairs_ok), "inliers": int(st.n_inliers)},
           "kernel_ms_per_pass": kernel_ms / steps, "call_ms_per_pass_incl_host_prepare_and_transfers": total_ms / steps,
           "wall_ms_per_pass_python": dt / steps * 1e3, "host_prepare_ms": st.host_prepare_ms}
    if cpu:
        try:
            from tests import _geofilter_cases as gc, _oracle
            if _oracle.have_ref_geofilter():
                m = min(cpu_pairs, n_pairs)
                sub = dict(xI=tv["xI"][:n * m], xJ=tv["xJ"][:n * m], start=tv["start"][:m + 1], wh=tv["wh"][:m])
                if model == "e":
                    ref_fn = lambda d: _oracle.ref_geofilter_e(d, K[:len(d["wh"])])   # noqa: E731
                else:
                    ref_fn = _oracle.ref_geofilter_h if model == "h" else _oracle.ref_geofilter
                ref_fn(dict(xI=sub["xI"][:n * 64], xJ=sub["xJ"][:n * 64], start=sub["start"][:65], wh=sub["wh"][:64]))
                ref = _at_least_a_second(ref_fn, sub)
                rec["cpu_baseline"] = {"value": m / ref["seconds"], "unit": "image pairs/s", "cores": os.cpu_count(), "kind": "reference",
                                       "sample": f"the first {m} pairs of the same set in {ref['seconds']:.2f} s (mean of {ref['repetitions']} runs; ACKernelAdaptor<"
                                                 f"{'FourPointSolver, AsymmetricError' if model == 'h' else 'FivePointSolver, EpipolarDistanceError (ACKernelAdaptorEssential)' if model == 'e' else 'SevenPointSolver, EpipolarDistanceError'}> + ACRANSAC, OpenMP over the pairs)"}
                rec["gpu_over_cpu"] = rec["value"] / rec["cpu_baseline"]["value"]   # whole call against whole call
                differing, rep = gc.compare(sub["start"], ref, mask[:n * m], res["ok"][:m], res["F"][:m], res["precision_robust"][:m], res["nfa"][:m])
                rec["parity"] = dict(rep, pairs_differing_between_two_builds_of_the_reference=REF_VS_REF.get(model),
                                     policy="identical inlier sets (then NFA, precision equal and F equal to 1e-6 - asserted); pairs_differing = "
                                                 "pairs whose decisive residual is within rounding of a histogram edge (DESIGN.md)")
        except Exception as e:
            rec["cpu_baseline"] = {"value": None, "kind": "reference", "sample": f"failed: {e!r}"}
    return rec


def geofilter_other_models_record(device=0, n_pairs=20000, n=250, cpu=True, cpu_pairs=2000):
    """The remaining -g models of main_GeometricFilter on the calibrated "f" set (one pass each): "a" / "u" = GeometricFilter_ESphericalMatrix_AC_
    Angular<false | true> (a-contrario stage; E_ACRobust_Angular.hpp), "o" = GeometricFilter_EOMatrix_RA (Eo_Robust.hpp). Per model: whole-call and
    kernel rates, the compiled reference on the first cpu_pairs pairs, parity with it (policy of the other records; "o" is closed form and
    bit-identical on identical inputs - tests/test_geofilter_ortho.py - here the inputs are numpy's bearing vectors, equal to 1e-16)."""
    import numpy as np
    from openmvg_amd import geofilter, synth
    tv = synth.two_view_matches_bulk(n_pairs, n=n, seed=0x6E0F)
    K = synth.two_view_calibration(tv)
    bI, bJ = geofilter.pinhole_bearings(K[0, 0], tv["xI"]), geofilter.pinhole_bearings(K[0, 1], tv["xJ"])   # (one calibration for the whole set)
    out = {}
    for model in ("a", "u", "o"):
        try:
            if model == "o":
                fun = geofilter.GeometricFilter_EOMatrix_RA(2.0, 1024)
                hI, hJ = np.ascontiguousarray(bI[:, :2] / bI[:, 2:3]), np.ascontiguousarray(bJ[:, :2] / bJ[:, 2:3])
                prec = np.full(n_pairs, (4.0 / K[0, 0, 0, 0] + 4.0 / K[0, 1, 0, 0]) / 2.0)
                run = lambda m: geofilter.filter_pairs_ortho_prepared(hI[:n * m], hJ[:n * m], tv["start"][:m + 1], tv["wh"][:m], prec[:m], fun, device)   # noqa: E731
            else:
                fun = geofilter.GeometricFilter_ESphericalMatrix_AC_Angular(4.0, 2048, model == "u")
                run = lambda m: geofilter.filter_pairs_angular(bI[:n * m], bJ[:n * m], tv["start"][:m + 1], fun, device)   # noqa: E731
            run(min(512, n_pairs))
            mask, res, st = run(n_pairs)
            rec = {"metric": "image pairs/s (a-contrario " + {"a": "angular essential, eight-point", "u": "angular essential, three-point upright",
                                                               "o": "orthographic essential"}[model] + " filter of putative matches)",
                   "value": n_pairs / (st.total_ms * 1e-3), "unit": "image pairs/s (whole call: host preparation, transfers, kernels)", "dtype": "f64",
                   "image_pairs_per_s_kernel_time": n_pairs / (st.kernel_ms * 1e-3), "kernel_ms_per_pass": st.kernel_ms,
                   "config": {"workload": f"{n_pairs} image pairs x {n} putative matches (the calibrated fundamental-matrix set)", "pairs_accepted": int(st.n_pairs_ok),
                              "iterations": int(st.n_iterations)}}
            if cpu:
                from tests import _geofilter_cases as gc, _oracle
                if _oracle.have_ref_geofilter():
                    m = min(cpu_pairs, n_pairs)
                    if model == "o":
                        sub = dict(xI=tv["xI"][:n * m], xJ=tv["xJ"][:n * m], start=tv["start"][:m + 1], wh=tv["wh"][:m])
                        ref = _at_least_a_second(lambda d: _oracle.ref_geofilter_eo(d, K[:m], precision=2.0, max_iterations=1024), sub)
                    else:
                        ref = _at_least_a_second(lambda d: _oracle.ref_geofilter_angular(bI[:n * m], bJ[:n * m], tv["start"][:m + 1], 4.0, 2048, upright=(model == "u")), None)
                    rec["cpu_baseline"] = {"value": m / ref["seconds"], "unit": "image pairs/s", "cores": os.cpu_count(), "kind": "reference",
                                           "sample": f"the first {m} pairs of the same set in {ref['seconds']:.3f} s (mean of {ref['repetitions']} runs; the functor's ACRANSAC stage, OpenMP over the pairs)"}
                    rec["gpu_over_cpu"] = rec["value"] / rec["cpu_baseline"]["value"]
                    differing, rep = gc.compare(tv["start"][:m + 1], ref, mask[:n * m], res["ok"][:m], res["F"][:m], res["precision_robust"][:m], res["nfa"][:m])
                    rec["parity"] = dict(rep, policy="identical inlier sets (then NFA, precision equal and E equal to 1e-6 - asserted)")
            out[model] = rec
        except Exception as e:
            out[model] = {"status": f"failed: {e!r}"}
    return out


if __name__ == "__main__":
    print(json.dumps(geofilter_bench_record(cpu="--no-cpu" not in sys.argv)))


def guided_matching_bench_record(device=0, n_pairs=2000, n=2000, steps=3, cpu=True, cpu_pairs=96):
    """The functors' second stage (guided_matching.hpp:178-227) on the device: n_pairs image pairs of n SIFT-like features per image under a
    fundamental matrix (4 px bound, ratio 0.8), with the reference's own template timed beside it on a sample of the pairs (one pair per host
    thread, as ImageCollectionGeometricFilter's OpenMP loop runs it) and the lists of that sample compared entry by entry."""
    import numpy as np
    from openmvg_amd import geofilter
    from tests.test_guided_matching import _pair
    rng = np.random.default_rng(11)
    base = [_pair(rng, n // 2, n - n // 2, n - n // 2, 0) for _ in range(8)]   # eight distinct pairs, cycled
    feats, descs, pairs, models = [], [], [], []
    for xi, di, xj, dj, M in base:
        feats += [xi, xj]; descs += [di, dj]
    for p in range(n_pairs):
        pairs.append((2 * (p % 8), 2 * (p % 8) + 1)); models.append(base[p % 8][4])
    prec = np.full(n_pairs, 4.0)
    geofilter.guided_matching(feats, descs, pairs[:64], models[:64], prec[:64], 0.8, 0, device)   # warm-up
    kernel_ms = total_ms = 0.0
    tests = passed = 0
    for _ in range(steps):
        res, st = geofilter.guided_matching(feats, descs, pairs, models, prec, 0.8, 0, device)
        kernel_ms += st.kernel_ms; total_ms += st.total_ms
        tests += int(st.n_geometric_tests); passed += int(st.n_geometric_passed)
    # VALU issue of the kernel per geometric test: 10 vector instructions in the loop body (5 fp64 operations, 3 compares, 2 mask operations; ISA of the
    # shipped kernel) - the descriptor stage of the 0.4 % that pass runs for the whole wave and is the larger half of the time
    valu_per_test = 10.0
    peak = 39.3216   # T lane-operations/s: 1 024 SIMDs x 16 lanes/clock x 2.4 GHz (full-rate fp64)
    ach = tests * valu_per_test / (kernel_ms * 1e-3) / 1e12
    rec = {"metric": "image pairs/s (guided matching: geometric bound + descriptor ratio, fundamental matrix)", "value": n_pairs * steps / (total_ms * 1e-3),
           "unit": "image pairs/s (whole call: uploads of positions and descriptors, kernels, compaction, read-back)", "dtype": "f64 predicate, u8 descriptors (exact int32)",
           "image_pairs_per_s_kernel_time": n_pairs * steps / (kernel_ms * 1e-3), "geometric_tests_per_s": tests / (kernel_ms * 1e-3),
           "fraction_of_tests_passed": passed / max(tests, 1),
           "roofline": {"bound": "valu", "achieved": ach, "peak": peak, "unit": "T lane-ops/s", "frac": ach / peak, "traffic": None,
                        "kernel": "guided_match_kernel<0, 32>", "mean_launch_ms": kernel_ms / steps,
                        "note": f"{valu_per_test:.0f} vector instructions per geometric test x tests/s against the full-rate VALU peak; the descriptor stage is not counted"},
           "config": {"workload": f"{n_pairs} image pairs x {n} x {n} features (half of them true correspondences under F, 1 px noise; 128-byte descriptors), "
                                  "bound 4 px, distance ratio 0.8", "matches": int(st.n_matches)}}
    if cpu:
        try:
            from concurrent.futures import ThreadPoolExecutor
            from tests import _oracle
            if _oracle.have_ref_geofilter():
                workers = min(os.cpu_count() or 1, n_pairs)
                m = max(cpu_pairs, workers)   # at least one pair per host thread, and as many rounds of that as a second of wall time takes
                def one(p):
                    I, J = pairs[p % n_pairs]
                    return _oracle.ref_guided_match(0, models[p % n_pairs], feats[I], descs[I], feats[J], descs[J], 16.0, 0.8 * 0.8)
                one(0)
                lists, secs, done = [], 0.0, 0
                with ThreadPoolExecutor(max_workers=workers) as ex:
                    while secs < 1.0 and done < 200 * m:
                        t0 = time.perf_counter()
                        part = list(ex.map(one, range(done, done + m)))
                        secs += time.perf_counter() - t0
                        if not lists:
                            lists = part
                        done += m
                rec["cpu_baseline"] = {"value": done / secs, "unit": "image pairs/s", "cores": workers, "kind": "reference",
                                       "sample": f"{done} pairs in {secs:.2f} s (geometry_aware::GuidedMatching<Mat3, EpipolarDistanceError> on SIFT_Regions, one pair per thread)"}
                rec["gpu_over_cpu"] = rec["value"] / rec["cpu_baseline"]["value"]
                same = all(np.array_equal(lists[p], res.get(pairs[p], np.zeros((0, 2), np.uint32))) for p in range(min(m, 8)))   # (pairs cycle with period 8)
                rec["parity"] = {"pairs_checked": m, "matches_checked": int(sum(len(v) for v in lists)), "identical": bool(same),
                                 "against": "the reference template's lists of the same pairs (same run)"}
        except Exception as e:
            rec["cpu_baseline"] = {"value": None, "kind": "reference", "sample": f"failed: {e!r}"}
    return rec
