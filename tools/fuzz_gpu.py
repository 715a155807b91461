"""Randomised parity campaign on the MI355X against the COMPILED reference (oracle/_ref: the reference's own Matcher_Regions and
Bundle_Adjustment_Ceres::Adjust): test infrastructure, not product.

    python tools/fuzz_gpu.py match <seconds> [seed]     ragged image sets (0 .. 6 000 descriptors per image, duplicated rows inside
                                                         and across images, every ratio the adapters pass) through mvgx_match_*; the
                                                         lists must EQUAL the reference's (regions_matcher.hpp:171-205)
    python tools/fuzz_gpu.py other <seconds> [seed]     the Hamming / float L2 / 144-byte uint8 L2 matchers likewise
    python tools/fuzz_gpu.py ba <seconds> [seed]        random scenes (3 .. 120 views, every camera model, shared intrinsics, track
                                                         lengths from 2 to 40 so that the usual groups, the wide groups and the
                                                         record-based path meet in one scene, outliers under the Huber loss, every
                                                         Optimize_Options combination) through mvgx_ba_*; final RMSE within 1e-6 of
                                                         Ceres' (north_star's tolerance), differences above 1e-9 listed
Prints one line per mismatch and a summary line; exit code 1 on any mismatch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from openmvg_amd import ba, matching, synth
from tests import _oracle


def fuzz_match(seconds, seed):
    rng = np.random.default_rng(seed)
    t0 = time.time(); n_cases = 0; bad = 0; n_pairs = 0; n_matches = 0
    pools = [list(range(0, 40)), list(range(500, 530)), [1023, 1024, 1025, 2047, 2048, 2049], list(range(900, 3100, 37)), [4095, 4096, 4097, 6000]]
    while time.time() - t0 < seconds:
        k = int(rng.integers(2, 7))
        sizes = [int(rng.choice(pools[int(rng.integers(0, len(pools)))])) for _ in range(k)]
        mode = int(rng.integers(0, 3))
        s = int(rng.integers(1 << 30))
        if mode == 0:
            imgs = synth.random_descriptors(k, sizes, seed=s)
        else:
            imgs = synth.image_descriptors(k, n_desc=max(sizes + [1]), seed=s)
            imgs = [d[:n] for d, n in zip(imgs, sizes)]
        if mode == 2 and min(sizes[0], sizes[1]) > 0:   # the same rows in two images and twice inside one image (ties for the first place)
            m = min(sizes[0], sizes[1]) // 2 + 1
            imgs[1] = imgs[1].copy(); imgs[1][:m] = imgs[0][:m]
            if sizes[1] > 3:
                imgs[1][1] = imgs[1][3]
        if rng.random() < 0.2 and sizes[0] > 0:          # extreme bytes
            imgs[0] = imgs[0].copy(); imgs[0][: max(1, sizes[0] // 4)] = rng.choice([0, 255], size=(max(1, sizes[0] // 4), 128)).astype(np.uint8)
        pairs = np.array([(i, j) for i in range(k) for j in range(i + 1, k)], np.uint32)
        if rng.random() < 0.3:
            pairs = pairs[:, ::-1].copy()
        ratio = float(rng.choice([0.6, 0.8, 0.8, 0.95, 1.0]))
        bp = int(rng.choice([0, 0, 1, 3]))
        ref = _oracle.ref_matcher_regions_match(imgs, pairs, ratio)
        ctx = matching.MatchContext(0)
        try:
            if bp:
                ctx.set_option("batch_pairs", bp)
            ctx.set_regions(imgs)
            r = np.float32(ratio)
            _, off, ij = ctx.run(pairs, r * r)
        finally:
            ctx.close()
        got = _oracle.offsets_to_dict(pairs, off, ij)
        ok = set(got) == set(ref) and all(np.array_equal(got[q], ref[q]) for q in ref)
        n_cases += 1; n_pairs += len(pairs); n_matches += int(sum(len(v) for v in ref.values()))
        if not ok:
            bad += 1
            print("MISMATCH match", sizes, mode, s, ratio, bp, flush=True)
    print(f"match: cases {n_cases} image pairs {n_pairs} matches {n_matches} mismatches {bad} ({time.time() - t0:.0f} s, seed {seed})", flush=True)
    return bad


def fuzz_other(seconds, seed):
    """the other instantiations of ArrayMatcherBruteForce (mvgx_bruteforce.hip): 64-byte binary rows under Hamming, 64-float rows under L2<float>
    (the reference's summation order: lists EQUAL), 144-byte uint8 rows under L2 - ragged sets, repeated rows, against the reference's own matcher"""
    rng = np.random.default_rng(seed)
    t0 = time.time(); n = [0, 0, 0]; bad = 0; n_matches = 0
    pools = [list(range(0, 20)), list(range(250, 270)), [63, 64, 65, 255, 256, 257, 1023, 1024, 1025], list(range(300, 2600, 53))]
    while time.time() - t0 < seconds:
        kind = int(rng.integers(0, 3))
        k = int(rng.integers(2, 6))
        sizes = [int(rng.choice(pools[int(rng.integers(0, len(pools)))])) for _ in range(k)]
        s = int(rng.integers(1 << 30))
        ratio = float(rng.choice([0.6, 0.8, 0.8, 0.95, 1.0]))
        bp = int(rng.choice([0, 0, 1, 3]))
        pairs = np.array([(i, j) for i in range(k) for j in range(k) if i != j], np.uint32)
        if kind == 0:
            imgs = synth.binary_descriptors(k, sizes, n_bytes=64, seed=s, flip_bits=int(rng.choice([8, 40, 120])))
            ctx = matching.HammingContext(0); arg = ratio; ref_fn = _oracle.ref_matcher_regions_match_binary64; L = 64
        elif kind == 1:
            imgs = synth.float_descriptors(k, sizes, dim=64, seed=s, noise=float(rng.choice([0.01, 0.05, 0.3])))
            ctx = matching.L2fContext(0); arg = np.float32(ratio) * np.float32(ratio); ref_fn = _oracle.ref_matcher_regions_match_float64; L = 64
        else:
            g = np.random.default_rng(s)
            imgs = [g.integers(0, 256, (m, 144), dtype=np.uint8) for m in sizes]
            for q in range(1, k):   # near-duplicates of the image before (matches exist), full byte range
                m = min(sizes[q], sizes[q - 1])
                if m:
                    imgs[q][:m] = np.clip(imgs[q - 1][:m].astype(np.int16) + g.integers(-9, 10, (m, 144)), 0, 255).astype(np.uint8)
            ctx = matching.L2u8Context(0); arg = np.float32(ratio) * np.float32(ratio); ref_fn = _oracle.ref_matcher_regions_match_liop144; L = 144
        if sizes[0] > 3 and rng.random() < 0.5:   # a row twice in one image: a tie for the first place
            imgs[0] = imgs[0].copy(); imgs[0][1] = imgs[0][3]
        try:
            if bp:
                ctx.set_option("batch_pairs", bp)
            ctx.set_regions(imgs, L)
            _, off, ij = ctx.run(pairs, arg)
        finally:
            ctx.close()
        ref = ref_fn(imgs, pairs, ratio)
        got = _oracle.offsets_to_dict(pairs, off, ij)
        ok = set(got) == set(ref) and all(np.array_equal(got[q], ref[q]) for q in ref)
        n[kind] += 1; n_matches += int(sum(len(v) for v in ref.values()))
        if not ok:
            bad += 1
            print("MISMATCH other", ["hamming", "l2_float", "l2_uint8_144"][kind], sizes, s, ratio, bp, flush=True)
    print(f"other matchers: cases hamming {n[0]} / float L2 {n[1]} / uint8-144 L2 {n[2]} matches {n_matches} mismatches {bad} ({time.time() - t0:.0f} s, seed {seed})", flush=True)
    return bad


def ba_cases(seed):
    """the campaign's scenes, a deterministic sequence per seed: (tag, scene, intrinsics_opt, extrinsics_opt, structure_opt, track lengths)"""
    rng = np.random.default_rng(seed)
    small = os.environ.get("FUZZ_SMALL") == "1"   # (a run under the CPU emulation)
    while True:
        model = int(rng.choice([1, 2, 3, 3, 4, 5, 7]))
        n_cams = int(rng.choice([5, 8, 20] if small else [3, 5, 8, 12, 20, 33, 60, 120]))
        n_pts = int(rng.choice([40, 120] if small else [40, 200, 900, 3000, 9000]))
        s = int(rng.integers(1 << 30))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            lens = np.full(n_pts, int(rng.integers(2, 11)), np.int64)
        elif kind == 1:
            lens = np.full(n_pts, int(rng.integers(11, 17)), np.int64)
        elif kind == 2:
            lens = synth.geometric_track_lengths(n_pts, mean=float(rng.choice([4.0, 6.0, 9.0])), lo=2, hi=40, seed=s & 0xFFFF)
        else:
            lens = rng.integers(2, 41, size=n_pts)
        lens = np.maximum(np.minimum(lens, n_cams), 2)
        groups = int(rng.integers(1, min(4, n_cams) + 1))
        outliers = float(rng.choice([0, 0, 0.02, 0.1])); rings = int(rng.choice([1, 2, 4]))
        iopt = int(rng.choice([14, 14, 1, 2, 6, 8, 10])) if model != 7 else 14
        eopt = int(rng.choice([6, 6, 2, 4, 1]))
        sopt = int(rng.choice([1, 1, 1, 0]))
        if model == 1 and iopt == 8:
            iopt = 14   # "distortion only" on a camera without distortion: every parameter constant through a SubsetParameterization - Ceres aborts
                        # (parameter_block.h:181 CHECK), the reference TU with it
        if eopt == 1 and sopt == 0 and iopt == 1:
            continue    # nothing to adjust
        sc = synth.ba_scene(n_cams=n_cams, n_points=n_pts, track_lens=lens, model=model, n_intr_groups=groups, seed=s, outlier_frac=outliers, n_rings=rings)
        yield (model, n_cams, n_pts, kind, groups, s, iopt, eopt, sopt), sc, iopt, eopt, sopt, lens


def adjust_both(sc, iopt, eopt, sopt, threads=8, max_iterations=0):
    """the caller's view on both sides: Adjust() with the reference's write-back rules (sfm_data_BA_ceres.cpp:527-568: ADJUST_ROTATION keeps
    the old centre), then the RMSE of the scene as it was left. Returns (ok, summary, rmse after, reference rc, reference stats)."""
    sc2 = dict(sc)
    adj = ba.Bundle_Adjustment_HIP()
    if max_iterations:
        adj.ceres_options().max_num_iterations_ = max_iterations
    ok = adj.Adjust(sc2, ba.Optimize_Options(iopt, eopt, sopt))
    c = ba.BaContext(sc2); _, rmse_after = c.evaluate(); c.close()
    rc, st, *_ = _oracle.ref_ba_adjust(sc, intrinsics_opt=iopt, extrinsics_opt=eopt, structure_opt=sopt, num_threads=threads, max_iterations=max_iterations)
    return ok, adj.summary, rmse_after, rc, st


def fuzz_ba(seconds, seed):
    """A difference counts as a mismatch when it exceeds 1e-6 (north_star) relative to max(1, RMSE) AND the reference reproduces its own figure at
    another thread count (where it does not, the case is listed as "unstable" with the three figures and not counted) AND the two sides already
    differ after five iterations (a long solve whose ends are apart but whose first five iterations agree to 1e-6 is listed as "slow")."""
    t0 = time.time(); n = 0; bad = 0; worst = 0.0; above = 0; wandering = 0; slow = 0; worst5 = 0.0; routes = np.zeros(3, np.int64)
    for tag, sc, iopt, eopt, sopt, lens in ba_cases(seed):
        if time.time() - t0 >= seconds:
            break
        try:
            ok, r, rmse_after, rc, st = adjust_both(sc, iopt, eopt, sopt)
        except Exception as e:   # noqa: BLE001
            print("EXC ba", *tag, repr(e)[:300], flush=True)
            bad += 1; n += 1
            continue
        n += 1
        routes += [int((lens <= 10).sum()), int(((lens > 10) & (lens <= 16)).sum()), int((lens > 16).sum())]
        diff = abs(rmse_after - float(st[1])) / max(1.0, float(st[1]))
        if rc != 0 or bool(st[3]) != bool(ok):
            bad += 1
            print("RC ba", *tag, "ok", ok, "rc", rc, st[3], flush=True)
        elif diff > 1e-6:
            # the reference against ITSELF at another thread count: Ceres' sums depend on it, and a trajectory that rounding separates (a solve far
            # from a minimum: outliers with constant centres, max_num_iterations reached) is not reproducible by the reference either
            rc1, st1, *_ = _oracle.ref_ba_adjust(sc, intrinsics_opt=iopt, extrinsics_opt=eopt, structure_opt=sopt, num_threads=1)
            own = abs(float(st1[1]) - float(st[1])) / max(1.0, float(st[1]))
            if own > 1e-7:
                wandering += 1
                print("unstable ba", *tag, "rmse", rmse_after, "reference 8 threads", float(st[1]), "1 thread", float(st1[1]), "iterations", r.num_iterations if r is not None else -1, flush=True)
            else:
                # a long, slowly converging solve (gross outliers under the Huber loss, a solve that runs into max_num_iterations): the function-tolerance
                # test fires iterations apart for different summation orders (tests/test_ba_gpu.py::test_huber_plateau_scene_...: the reference's own
                # builds 4e-4 apart). What must agree is the trajectory while rounding has not separated it: both sides again, five iterations
                ok5, r5, rmse5, rc5, st5 = adjust_both(sc, iopt, eopt, sopt, max_iterations=5)
                d5 = abs(rmse5 - float(st5[1])) / max(1.0, float(st5[1]))
                if d5 <= 1e-6:
                    slow += 1; worst5 = max(worst5, d5)
                    print("slow ba", *tag, "rmse", rmse_after, float(st[1]), "iterations", r.num_iterations if r is not None else -1, "after five iterations", rmse5, float(st5[1]), flush=True)
                else:
                    bad += 1
                    print("DIFF ba", *tag, "rmse", rmse_after, float(st[1]), "iterations", r.num_iterations if r is not None else -1, "after five iterations", rmse5, float(st5[1]), flush=True)
        else:
            worst = max(worst, diff)
            above += diff > 1e-9
    print(f"ba: cases {n} bad {bad} cases the reference does not reproduce itself (listed, not counted) {wandering} slow solves that stop apart but agree after five iterations (listed, not counted) {slow} (largest relative difference there {worst5:.3e}) largest relative |RMSE - Ceres| of the rest {worst:.3e} "
          f"cases above 1e-9: {above} points by route (usual / wide / records) {routes.tolist()} ({time.time() - t0:.0f} s, seed {seed})", flush=True)
    return bad


def replay_ba(seed, wanted):
    """ba-replay <seed> <s,s,...>: the named cases of a campaign again, against the compiled reference AND the restatement (oracle/ba_oracle.cpp, the
    same algorithm in Jets on the CPU): device = restatement says the device follows the algorithm and the reference's trajectory separated by rounding"""
    wanted = set(wanted)
    for tag, sc, iopt, eopt, sopt, lens in ba_cases(seed):
        if tag[5] not in wanted:
            continue
        wanted.discard(tag[5])
        masks = ba.bo.masks_for(sc, iopt, eopt, sopt)
        c = ba.BaContext(sc, **masks); s = c.solve(); c.close()
        rc, osum, *_ = _oracle.port_ba_solve(sc, **masks)
        line = {"case": tag, "device": (s.num_iterations, s.final_rmse), "restatement": (osum.num_iterations, osum.final_rmse)}
        for th in (1, 8):
            rc, st, *_ = _oracle.ref_ba_adjust(sc, intrinsics_opt=iopt, extrinsics_opt=eopt, structure_opt=sopt, num_threads=th)
            line[f"reference_{th}_threads_rmse_after_write_back"] = float(st[1])
        ok, r, rmse_after, rc, st = adjust_both(sc, iopt, eopt, sopt)
        line["device_rmse_after_write_back"] = rmse_after
        print(line, flush=True)
        if not wanted:
            break


if __name__ == "__main__":
    what = sys.argv[1]
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    if what == "ba-replay":   # ba-replay <seed> <s,s,...>
        replay_ba(int(sys.argv[2]), [int(x) for x in sys.argv[3].split(",")])
        sys.exit(0)
    rc = {"match": fuzz_match, "other": fuzz_other, "ba": fuzz_ba}[what](seconds, seed)
    sys.exit(1 if rc else 0)
