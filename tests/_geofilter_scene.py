"""An in-memory image collection for the container-level geometric-filter tests: images with feature positions, putative matches
between consecutive images built from synth.two_view_matches (features shuffled, a few unmatched features added)."""
import numpy as np

from openmvg_amd import synth


def collection(n_pairs=12, seed=5, n_max=120, size=(1000, 1000), homography=False, **kw):
    gen = synth.two_view_homography_matches if homography else synth.two_view_matches
    tv = gen(n_pairs, seed=seed, n_max=n_max, tiny_frac=0.1, sizes=(size,), **kw)
    start = tv["start"].astype(np.int64)
    feats, putative = [], {}
    for p in range(n_pairs):   # images 2 p and 2 p + 1
        n = int(start[p + 1] - start[p])
        rng = np.random.default_rng(seed * 1000 + p)
        extra = 5
        perm_i, perm_j = rng.permutation(n + extra)[:n], rng.permutation(n + extra)[:n]
        fi = rng.uniform(0, size[0], (n + extra, 2)).astype(np.float32); fj = rng.uniform(0, size[0], (n + extra, 2)).astype(np.float32)
        fi[perm_i] = tv["xI"][start[p]:start[p + 1]].astype(np.float32); fj[perm_j] = tv["xJ"][start[p]:start[p + 1]].astype(np.float32)
        feats += [fi, fj]
        putative[(2 * p, 2 * p + 1)] = np.stack([perm_i, perm_j], 1).astype(np.uint32)
    wh = np.tile(np.asarray(size, np.uint32), (2 * n_pairs, 1))
    return feats, wh, putative
