#!/usr/bin/env python
"""Side record of bench.py: BRUTE_FORCE_HAMMING on AKAZE-like 64-byte binary descriptors on one MI355X.

Workload: 300 images x 2000 descriptors, exhaustive pairs (44 850 image pairs = 1.79e11 descriptor pairs). The kernel is
VALU-bound: 2 x 16 + 3 = 35 32-bit integer lane operations per descriptor pair (xor + popcount-accumulate per dword, packed
key, min, max/min); peak = 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane-ops/s (MI355X_MICROARCH.md)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

VALU_LANE_OPS_PEAK = 256 * 4 * 16 * 2.4e9
LANE_OPS_PER_DESC_PAIR = 35.0


def _cpu_leg(rec, ctx, imgs, pairs, n_desc, run_ratio, cpu_seconds, ref_fn, port_fn, label):
    """The reference's Matcher_Regions on a bounded random sample of the same pair list (cpu_baseline), and the device lists of
    those very pairs - same context, same resident descriptors as the timed region - compared entry by entry (parity)."""
    try:
        from tests import _oracle
        kind = "reference" if _oracle.have_ref_match() else "port"
        fn = ref_fn(_oracle) if kind == "reference" else (lambda d, p, r: _oracle.offsets_to_dict(p, *port_fn(_oracle)(d, p, r)))
        order = np.random.default_rng(1).permutation(len(pairs))
        fn(imgs, pairs[order[:8]], 0.8)
        t0 = time.perf_counter(); fn(imgs, pairs[order[8:40]], 0.8)
        per = max((time.perf_counter() - t0) / 32.0, 1e-5)
        n = int(max(32, min(len(pairs), cpu_seconds / per)))
        sample = np.ascontiguousarray(pairs[order[:n]])
        t0 = time.perf_counter(); cpu_lists = fn(imgs, sample, 0.8); cdt = time.perf_counter() - t0
        rec["cpu_baseline"] = {"value": n * n_desc * n_desc / cdt, "unit": "descriptor pairs/s", "cores": os.cpu_count(), "kind": kind,
                               "sample": f"{n} random image pairs of the same set in {cdt:.1f} s (Matcher_Regions, {label})"}
        rec["gpu_over_cpu"] = rec["value"] / rec["cpu_baseline"]["value"]
        _, soff, sij = ctx.run(sample, run_ratio)
        gpu = _oracle.offsets_to_dict(sample, soff, sij)
        same = set(gpu) == set(cpu_lists) and all(np.array_equal(gpu[k], cpu_lists[k]) for k in gpu)
        rec["parity"] = {"pairs_checked": int(n), "non_empty_pairs": int(len(cpu_lists)),
                         "matches_checked": int(sum(len(v) for v in cpu_lists.values())), "identical": bool(same),
                         "against": "cpu_baseline lists (same run, same pairs)"}
    except Exception as e:
        rec["cpu_baseline"] = {"value": None, "kind": "port", "sample": f"failed: {e!r}"}


def hamming_bench_record(device=0, n_images=300, n_desc=2000, steps=3, cpu_seconds=8.0, cpu=True):
    from openmvg_amd import matching, synth
    imgs = synth.binary_descriptors(n_images, n_desc, seed=0xA4A2E)
    pairs = matching.exhaustive_pairs_array(n_images)
    ctx = matching.HammingContext(device)
    ctx.set_regions(imgs, 64)
    ctx.run(pairs[:2000], 0.8)   # warm-up
    kernel_ms = 0.0; launches = 0; desc_pairs = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        st, off, _ = ctx.run(pairs, 0.8)
        kernel_ms += st.kernel_ms; launches += int(st.n_kernel_launches); desc_pairs += int(st.n_desc_pairs)
    dt = time.perf_counter() - t0
    ach = desc_pairs * LANE_OPS_PER_DESC_PAIR / max(kernel_ms * 1e-3, 1e-12)
    rec = {
        "metric": "descriptor pairs/s (brute-force Hamming 2-NN + ratio matching)", "value": desc_pairs / dt,
        "unit": "descriptor pairs/s", "dtype": "u32 popcount",
        "config": {"workload": f"{n_images} images x {n_desc} 64-byte binary descriptors, exhaustive pairs ({len(pairs)} image pairs), ratio 0.8",
                   "matches": int(off[-1])},
        "ms_per_step": dt / steps * 1e3,
        "roofline": {"bound": "valu", "achieved": ach / 1e12, "peak": VALU_LANE_OPS_PEAK / 1e12, "unit": "T lane-ops/s",
                     "frac": ach / VALU_LANE_OPS_PEAK, "traffic": None, "kernel": "hamming_top2_ratio_kernel<16>",
                     "launches": launches, "mean_launch_ms": kernel_ms / max(launches, 1)},
    }
    if cpu:
        _cpu_leg(rec, ctx, imgs, pairs, n_desc, 0.8, cpu_seconds, lambda o: o.ref_matcher_regions_match_binary64,
                 lambda o: o.port_matcher_regions_match_hamming, "BRUTE_FORCE_HAMMING")
    ctx.close()
    return rec


PK_F32_OPS_PEAK = 256 * 4 * 16 * 2 * 2.4e9     # packed fp32, one operation per half and lane and cycle (no FMA allowed here)
F32_OPS_PER_DESC_PAIR = 3.0 * 64                # sub, mul, add per element, in the reference's order


def l2f_bench_record(device=0, n_images=200, n_desc=2000, steps=3, cpu_seconds=8.0, cpu=True):
    """BRUTE_FORCE_L2 on AKAZE_Float_Regions-like 64-float descriptors: 200 images x 2000, exhaustive pairs."""
    from openmvg_amd import matching, synth
    imgs = synth.float_descriptors(n_images, n_desc, seed=0xF10A7)
    pairs = matching.exhaustive_pairs_array(n_images)
    rsq = np.float32(0.8) * np.float32(0.8)
    ctx = matching.L2fContext(device)
    ctx.set_regions(imgs, 64)
    ctx.run(pairs[:1000], rsq)
    kernel_ms = 0.0; launches = 0; desc_pairs = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        st, off, _ = ctx.run(pairs, rsq)
        kernel_ms += st.kernel_ms; launches += int(st.n_kernel_launches); desc_pairs += int(st.n_desc_pairs)
    dt = time.perf_counter() - t0
    ach = desc_pairs * F32_OPS_PER_DESC_PAIR / max(kernel_ms * 1e-3, 1e-12)
    rec = {
        "metric": "descriptor pairs/s (brute-force L2<float> 2-NN + ratio matching, reference summation order)",
        "value": desc_pairs / dt, "unit": "descriptor pairs/s", "dtype": "f32 (separate mul/add, no contraction)",
        "config": {"workload": f"{n_images} images x {n_desc} 64-float descriptors, exhaustive pairs ({len(pairs)} image pairs), ratio 0.8",
                   "matches": int(off[-1])},
        "ms_per_step": dt / steps * 1e3,
        "roofline": {"bound": "valu", "achieved": ach / 1e12, "peak": PK_F32_OPS_PEAK / 1e12, "unit": "T fp32 ops/s",
                     "frac": ach / PK_F32_OPS_PEAK, "traffic": None, "kernel": "l2f_top2_ratio_kernel<64>",
                     "launches": launches, "mean_launch_ms": kernel_ms / max(launches, 1)},
    }
    if cpu:
        _cpu_leg(rec, ctx, imgs, pairs, n_desc, rsq, cpu_seconds, lambda o: o.ref_matcher_regions_match_float64,
                 lambda o: o.port_matcher_regions_match_f32, "BRUTE_FORCE_L2, float")
    ctx.close()
    return rec


VALU_I32_OPS_PEAK = 256 * 4 * 16 * 2.4e9           # one integer lane-op per lane and cycle (v_dot4_u32_u8 counts as one)
L2U8_OPS_PER_DESC_PAIR_144 = 36 + 9               # 36 dot products + 9 other VALU per descriptor pair at 144 bytes


def liop_like_descriptors(n_images, n_desc, dim=144, seed=0x110B, window_factor=3.0, noise=6):
    """AKAZE-LIOP-like uint8 descriptors with real correspondences: a world of uniform byte vectors, every image draws n_desc
    distinct world ids from a sliding window (neighbouring images share ~a third) and adds integer noise U{-noise..noise}
    per bin - the same construction as synth.image_descriptors, at 144 bytes, so that the ratio test accepts matches."""
    n_world = max(20 * n_images, int(window_factor * n_desc) + 1)
    world = np.random.default_rng(seed).integers(0, 256, size=(n_world, dim), dtype=np.uint8)
    win = min(n_world, int(window_factor * n_desc))
    out = []
    for k in range(n_images):
        rng = np.random.default_rng(seed + 1 + k)
        start = int((n_world - win) * (k / max(1, n_images - 1))) if n_images > 1 else 0
        ids = start + rng.choice(win, size=n_desc, replace=False)
        d = world[ids].astype(np.int16) + rng.integers(-noise, noise + 1, size=(n_desc, dim), dtype=np.int16)
        out.append(np.clip(d, 0, 255).astype(np.uint8))
    return out


def l2u8_bench_record(device=0, n_images=200, n_desc=2000, steps=3, dim=144, cpu_seconds=8.0, cpu=True):
    """BRUTE_FORCE_L2 on AKAZE_Liop_Regions-like 144-byte uint8 descriptors (the third mvgx_bruteforce kernel): 200 images x 2000,
    exhaustive pairs, descriptors with true correspondences (the lists are non-empty: compaction and the D2H of the lists are
    inside the timed region). CPU leg: the reference's Matcher_Regions on AKAZE_Liop_Regions over a bounded sample of the pairs,
    and the device lists of those very pairs compared entry by entry."""
    from openmvg_amd import matching
    imgs = liop_like_descriptors(n_images, n_desc, dim)
    pairs = matching.exhaustive_pairs_array(n_images)
    rsq = np.float32(0.8) * np.float32(0.8)
    ctx = matching.L2u8Context(device)
    ctx.set_regions(imgs, dim)
    ctx.run(pairs[:1000], rsq)
    kernel_ms = 0.0; launches = 0; desc_pairs = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        st, off, _ = ctx.run(pairs, rsq)
        kernel_ms += st.kernel_ms; launches += int(st.n_kernel_launches); desc_pairs += int(st.n_desc_pairs)
    dt = time.perf_counter() - t0
    ach = desc_pairs * L2U8_OPS_PER_DESC_PAIR_144 / max(kernel_ms * 1e-3, 1e-12)
    rec = {"metric": "descriptor pairs/s (brute-force L2<uint8> 2-NN + ratio matching, 144-byte descriptors)",
           "value": desc_pairs / dt, "unit": "descriptor pairs/s", "dtype": "u8 (v_dot4_u32_u8, exact int32)",
           "config": {"workload": f"{n_images} images x {n_desc} {dim}-byte descriptors with true correspondences, exhaustive pairs "
                                  f"({len(pairs)} image pairs), ratio 0.8",
                      "matches": int(off[-1])},
           "ms_per_step": dt / steps * 1e3,
           "roofline": {"bound": "valu", "achieved": ach / 1e12, "peak": VALU_I32_OPS_PEAK / 1e12, "unit": "T lane-ops/s",
                        "frac": ach / VALU_I32_OPS_PEAK, "traffic": None, "kernel": "l2u8_top2_ratio_kernel<36>", "launches": launches,
                        "mean_launch_ms": kernel_ms / max(launches, 1)}}
    if cpu:
        _cpu_leg(rec, ctx, imgs, pairs, n_desc, rsq, cpu_seconds, lambda o: o.ref_matcher_regions_match_liop144,
                 lambda o: (lambda d, p, r: o.port_matcher_regions_match(d, p, r, dim=dim)), "BRUTE_FORCE_L2, AKAZE_Liop_Regions")
    ctx.close()
    return rec


if __name__ == "__main__":
    if "l2u8" in sys.argv:
        print(json.dumps(l2u8_bench_record(cpu="--no-cpu" not in sys.argv)))
        sys.exit(0)
    if "l2f" in sys.argv:
        print(json.dumps(l2f_bench_record(cpu="--no-cpu" not in sys.argv)))
        sys.exit(0)
    print(json.dumps(hamming_bench_record(cpu="--no-cpu" not in sys.argv)))
