"""On-disk formats either side of the matching hot path, cereal-free, so that the accelerators can be
driven from an openMVG matches directory:

  <image>.desc   binary descriptors   features/descriptor.hpp:182-226 (loadDescsFromBinFile / saveDescsToBinFile):
                 std::size_t count, then count x 128 bytes (Descriptor<unsigned char, 128>)
  <image>.feat   text features        features/feature.hpp:58-95, feature.cpp:29-37,69-79 (SIOPointFeature stream operators,
                 loadFeatsFromFile features/feature_container.hpp): one "x y scale orientation" line per feature
  matches.*.txt  putative matches     matching/indMatch_utils.cpp:85-131 (Save, "txt" branch) / :28-83 (Load):
                 per pair "I J\n#matches\n" then one "i j" line per IndMatch (matching/indMatch.hpp:58-64)

  <scene>.baf    BA problem export    sfm/sfm_data_io_baf.hpp:38-147 (Save_BAF; the reference has no BAF reader): header counts,
                 one line per intrinsic (getParams()), per view (R column-major 3x3 + centre; identity / 0 when the pose
                 is missing), per landmark (X, #obs, then "id_intrinsic id_pose x y" per observation), plus
                 <scene>_imgList.txt ("filename id_intrinsic id_pose" per view)

Host-side plumbing only (numpy); the .bin variants of the reference are cereal archives and are out of scope (the cereal
submodule is absent from the reference tree).
"""
import numpy as np

DESC_DIM = 128


def load_desc_bin(path, dim=DESC_DIM, dtype=np.uint8):
    """-> (n, dim) array of `dtype`, C-contiguous. SIFT_Regions: (128, uint8) - what mvgx_match_set_regions takes;
    AKAZE_Binary_Regions: (64, uint8); AKAZE_Float_Regions: (64, float32). The file is `std::size_t count` followed by
    count x dim elements of the descriptor's bin_type (descriptor.hpp:182-203)."""
    dtype = np.dtype(dtype)
    with open(path, "rb") as f:
        head = f.read(8)
        if len(head) != 8:
            raise ValueError(f"{path}: truncated header")
        n = int(np.frombuffer(head, "<u8")[0])
        data = np.frombuffer(f.read(n * dim * dtype.itemsize), dtype)
    if data.size != n * dim:
        raise ValueError(f"{path}: expected {n} x {dim} elements, found {data.size}")
    return data.reshape(n, dim).copy()


def save_desc_bin(path, desc, dtype=np.uint8):
    desc = np.ascontiguousarray(desc, dtype)
    with open(path, "wb") as f:
        f.write(np.array([desc.shape[0]], "<u8").tobytes())
        f.write(desc.tobytes())


def load_feat(path):
    """-> (n, 4) float32: x, y, scale, orientation."""
    a = np.loadtxt(path, dtype=np.float32, ndmin=2)
    return a.reshape(-1, 4) if a.size else np.zeros((0, 4), np.float32)


def save_feat(path, feats):
    """operator<< of SIOPointFeature: default ostream formatting of floats (6 significant digits, %g-like)."""
    with open(path, "w") as f:
        for x, y, s, o in np.asarray(feats, np.float32).reshape(-1, 4):
            f.write(f"{float(x):g} {float(y):g} {float(s):g} {float(o):g}\n")


def save_matches_txt(path, pairs, offsets, ij):
    """pairs (n_pairs x 2), offsets (n_pairs + 1), ij (n_matches x 2) as returned by MatchContext.run(): one block per
    NON-EMPTY pair (Matcher_Regions.cpp:99-102 inserts only those), in ascending (I, J) - the iteration order of the
    reference's std::map<Pair, IndMatches>."""
    pairs = np.asarray(pairs).reshape(-1, 2)
    order = np.lexsort((pairs[:, 1], pairs[:, 0]))
    ij = np.asarray(ij).reshape(-1, 2)
    with open(path, "w") as f:
        for k in order:
            lo, hi = int(offsets[k]), int(offsets[k + 1])
            if hi == lo:
                continue
            f.write(f"{int(pairs[k, 0])} {int(pairs[k, 1])}\n{hi - lo}\n")
            np.savetxt(f, ij[lo:hi], fmt="%d", delimiter=" ")


def load_matches_txt(path):
    """-> dict {(I, J): (n, 2) uint32}."""
    out = {}
    with open(path) as f:
        tok = f.read().split()
    p = 0
    while p + 3 <= len(tok):
        I, J, n = int(tok[p]), int(tok[p + 1]), int(tok[p + 2])
        p += 3
        out[(I, J)] = np.array(tok[p:p + 2 * n], np.uint32).reshape(n, 2)
        p += 2 * n
    return out


def _g(x):
    """default std::ostream formatting of a double / unsigned (precision 6, %g)."""
    return f"{float(x):g}"


def save_baf(path, poses, intrinsics, intr_model, points, obs_pose, obs_intr, obs_point, obs_xy, image_names=None,
             root_path=""):
    """Save_BAF (sfm/sfm_data_io_baf.hpp:38-147) for a flat BA scene (the layout of mvgx_ba_problem): poses n x 6
    (angle-axis, t = -R C; one view per pose, view id = pose id, using the intrinsic of its first observation),
    intrinsics n x 8 + model ids, points n x 3, observations. The reference iterates its std::unordered_map containers,
    so its line order inside a section is libstdc++'s bucket order; this writer emits ascending ids (the multiset of lines
    is what tests/test_io_cpu.py pins against the reference)."""
    from .ba_options import N_INTR_PARAMS
    poses = np.asarray(poses, np.float64).reshape(-1, 6)
    intrinsics = np.asarray(intrinsics, np.float64).reshape(-1, 8)
    points = np.asarray(points, np.float64).reshape(-1, 3)
    obs_pose = np.asarray(obs_pose, np.int64)
    obs_intr = np.asarray(obs_intr, np.int64)
    obs_point = np.asarray(obs_point, np.int64)
    obs_xy = np.asarray(obs_xy, np.float64).reshape(-1, 2)
    n_poses = len(poses)
    pose_intr = np.zeros(n_poses, np.int64)
    seen = np.zeros(n_poses, bool)
    for p, i in zip(obs_pose, obs_intr):
        if not seen[p]:
            seen[p] = True
            pose_intr[p] = i
    lines = [str(len(intrinsics)), str(n_poses), str(len(points))]
    for prm, model in zip(intrinsics, np.asarray(intr_model).reshape(-1)):
        n = N_INTR_PARAMS[int(model)]
        if int(model) == 7:   # Intrinsic_Spherical::getParams() is empty (Camera_Spherical.hpp)
            n = 0
        lines.append("".join(_g(v) + " " for v in prm[:n]))
    for p in poses:
        w = p[:3]
        th = float(np.linalg.norm(w))
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        if th > 0:
            R = np.eye(3) + (np.sin(th) / th) * K + ((1 - np.cos(th)) / (th * th)) * (K @ K)
        else:
            R = np.eye(3) + K
        C = -R.T @ p[3:6]
        lines.append("".join(_g(v) + " " for v in R.T.reshape(-1)) + "".join(_g(v) + " " for v in C))
    order = np.lexsort((obs_pose, obs_point))
    starts = np.searchsorted(obs_point[order], np.arange(len(points) + 1))
    for j, X in enumerate(points):
        ks = order[starts[j]:starts[j + 1]]
        line = "".join(_g(v) + " " for v in X) + f"{len(ks)} "
        for k in ks:
            line += f"{pose_intr[obs_pose[k]]} {obs_pose[k]} {_g(obs_xy[k, 0])} {_g(obs_xy[k, 1])} "
        lines.append(line)
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    import os
    stem, _ = os.path.splitext(path)
    with open(stem + "_imgList.txt", "w") as f:
        for v in range(n_poses):
            name = image_names[v] if image_names is not None else ""
            full = name if not root_path else (root_path.rstrip("/") + "/" + name)
            f.write(f"{full} {pose_intr[v]} {v}\n")


def match_directory(match_dir, stems, ratio=0.8, kind="sift", pairs=None, out_name="matches.putative.txt", device=-1):
    """File-level pipeline either side of the matching path (main_ComputeMatches without the cereal formats): reads
    `<match_dir>/<stem>.desc` for the ordered list `stems` (view id = position in the list, as main_ComputeMatches takes it
    from the views of sfm_data), matches the pairs (default: exhaustivePairs, Pair_Builder.hpp:25-33) on the device and
    writes `<match_dir>/<out_name>` in the reference's text format. kind: "sift" (uint8 x 128, BRUTE_FORCE_L2), "binary"
    (uint8 x 64, BRUTE_FORCE_HAMMING) or "float" (float32 x 64, BRUTE_FORCE_L2). Returns {(I, J): (n, 2) uint32}."""
    import os
    from . import matching
    spec = {"sift": (128, np.uint8, matching.Regions, matching.EMatcherType.BRUTE_FORCE_L2),
            "binary": (64, np.uint8, matching.Binary_Regions, matching.EMatcherType.BRUTE_FORCE_HAMMING),
            "float": (64, np.float32, matching.Float_Regions, matching.EMatcherType.BRUTE_FORCE_L2)}[kind]
    dim, dtype, cls, mtype = spec
    regions = {k: cls(load_desc_bin(os.path.join(match_dir, stem + ".desc"), dim, dtype)) for k, stem in enumerate(stems)}
    provider = matching.Regions_Provider(regions)
    if pairs is None:
        pairs = matching.exhaustivePairs(len(stems))
    out = matching.PairWiseMatches()
    matching.Matcher_Regions(ratio, mtype, device=device).Match(provider, pairs, out)
    keys = sorted(out)
    offsets = np.zeros(len(keys) + 1, np.uint64)
    for k, key in enumerate(keys):
        offsets[k + 1] = offsets[k] + len(out[key])
    ij = np.concatenate([out[k] for k in keys]) if keys else np.zeros((0, 2), np.uint32)
    save_matches_txt(os.path.join(match_dir, out_name), np.array(keys, np.uint32).reshape(-1, 2), offsets, ij)
    return dict(out)
