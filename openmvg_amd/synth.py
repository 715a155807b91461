"""Deterministic synthetic inputs of the shapes BASELINE.json names.

Descriptors: a "world" of RootSIFT-like vectors (128 iid Exp(1) draws, L1-normalise, sqrt, x512, clamp to
[0, 255] — mirrors the reference's sift_DescriptorExtractor.hpp:484-494 quantisation); image k samples distinct
world points from a sliding window (neighbouring images overlap, distant ones do not) and adds integer noise
U{-3..3} per bin. numpy's PCG64 replaces the std::mt19937_64 of the survey text; seeds are fixed here.
"""
import numpy as np


def world_descriptors(n_world, seed=0x5EED0000):
    rng = np.random.default_rng(seed)
    e = rng.standard_exponential((n_world, 128), dtype=np.float32)
    e /= e.sum(axis=1, keepdims=True)
    v = np.sqrt(e) * 512.0
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def image_descriptors(n_images, n_desc=2000, seed=0xC0FFEE00, world=None, window_factor=3.0):
    """Returns a list of n_images arrays (n_desc, 128) uint8."""
    n_world = 20 * n_images if world is None else world.shape[0]
    if world is None:
        world = world_descriptors(max(n_world, int(window_factor * n_desc) + 1))
        n_world = world.shape[0]
    win = min(n_world, int(window_factor * n_desc))
    out = []
    for k in range(n_images):
        rng = np.random.default_rng(seed + k)
        start = int((n_world - win) * (k / max(1, n_images - 1))) if n_images > 1 else 0
        ids = start + rng.choice(win, size=n_desc, replace=False)
        d = world[ids].astype(np.int16) + rng.integers(-3, 4, size=(n_desc, 128), dtype=np.int16)
        out.append(np.clip(d, 0, 255).astype(np.uint8))
    return out


def random_descriptors(n_images, n_desc, seed=1):
    """iid uniform bytes — worst case for norms/ranges, used by exactness tests."""
    rng = np.random.default_rng(seed)
    if np.isscalar(n_desc):
        n_desc = [int(n_desc)] * n_images
    return [rng.integers(0, 256, size=(int(n), 128), dtype=np.uint8) for n in n_desc]
