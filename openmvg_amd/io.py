"""On-disk formats either side of the matching hot path, cereal-free, so that the accelerators can be
driven from an openMVG matches directory:

  <image>.desc   binary descriptors   features/descriptor.hpp:182-226 (loadDescsFromBinFile / saveDescsToBinFile):
                 std::size_t count, then count x 128 bytes (Descriptor<unsigned char, 128>)
  <image>.feat   text features        features/feature.hpp:58-95, feature.cpp:29-37,69-79 (SIOPointFeature stream operators,
                 loadFeatsFromFile features/feature_container.hpp): one "x y scale orientation" line per feature
  matches.*.txt  putative matches     matching/indMatch_utils.cpp:85-131 (Save, "txt" branch) / :28-83 (Load):
                 per pair "I J\n#matches\n" then one "i j" line per IndMatch (matching/indMatch.hpp:58-64)

Host-side plumbing only (numpy); the .bin variants of the reference are cereal archives and are out of scope (the cereal
submodule is absent from the reference tree).
"""
import numpy as np

DESC_DIM = 128


def load_desc_bin(path, dim=DESC_DIM):
    """-> (n, dim) uint8, C-contiguous (what mvgx_match_set_regions takes)."""
    with open(path, "rb") as f:
        head = f.read(8)
        if len(head) != 8:
            raise ValueError(f"{path}: truncated header")
        n = int(np.frombuffer(head, "<u8")[0])
        data = np.frombuffer(f.read(n * dim), np.uint8)
    if data.size != n * dim:
        raise ValueError(f"{path}: expected {n} x {dim} bytes, found {data.size}")
    return data.reshape(n, dim).copy()


def save_desc_bin(path, desc):
    desc = np.ascontiguousarray(desc, np.uint8)
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
