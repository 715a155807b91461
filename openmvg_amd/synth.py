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


# ---------------------------------------------------------------------------------------------------------
# Bundle-adjustment scenes: cameras on rings looking at the origin (generalised
# NRealisticCamerasRing, reference multiview/test_data_sets.cpp:45-88: f = 1000, pp = (500, 500)), points in a cube,
# every point observed by `track_len` consecutive cameras; observations = exact projection + N(0, noise_px);
# initial state = ground truth perturbed (rotations N(0, rot_deg) angle-axis noise, centres and points N(0, 0.01)).
# ---------------------------------------------------------------------------------------------------------
CAM_PINHOLE, CAM_PINHOLE_RADIAL1, CAM_PINHOLE_RADIAL3, CAM_PINHOLE_BROWN, CAM_PINHOLE_FISHEYE, CAM_SPHERICAL = 1, 2, 3, 4, 5, 7
_NPARAM = {1: 3, 2: 4, 3: 6, 4: 8, 5: 7, 7: 0}


def _rodrigues(aa):
    """angle-axis (n,3) -> rotation matrices (n,3,3)."""
    aa = np.asarray(aa, np.float64).reshape(-1, 3)
    th = np.linalg.norm(aa, axis=1)
    safe = np.where(th > 1e-12, th, 1.0)
    k = aa / safe[:, None]
    Kx = np.zeros((len(aa), 3, 3))
    Kx[:, 0, 1] = -k[:, 2]; Kx[:, 0, 2] = k[:, 1]; Kx[:, 1, 0] = k[:, 2]
    Kx[:, 1, 2] = -k[:, 0]; Kx[:, 2, 0] = -k[:, 1]; Kx[:, 2, 1] = k[:, 0]
    s = np.where(th > 1e-12, np.sin(th), th)[:, None, None]
    c = np.where(th > 1e-12, 1 - np.cos(th), 0.0)[:, None, None]
    return np.eye(3)[None] + s * Kx + c * np.einsum("nij,njk->nik", Kx, Kx)


def _rotmat_to_aa(R):
    from scipy.spatial.transform import Rotation
    return Rotation.from_matrix(R).as_rotvec()


def project(model, intr, pose, X):
    """numpy restatement of the residual functors' projection (float64), vectorised over observations
    (sfm_data_BA_ceres_camera_functor.hpp: pinhole, radial K1/K3, Brown T2, fisheye, spherical)."""
    R = _rodrigues(pose[:, :3])
    p = np.einsum("nij,nj->ni", R, X) + pose[:, 3:6]
    if model == CAM_SPHERICAL:   # intr = {w, h}
        lon = np.arctan2(p[:, 0], p[:, 2])
        lat = np.arctan2(-p[:, 1], np.hypot(p[:, 0], p[:, 2]))
        size = np.maximum(intr[:, 0], intr[:, 1])
        return np.stack([lon / (2 * np.pi) * size + intr[:, 0] / 2, -lat / (2 * np.pi) * size + intr[:, 1] / 2], axis=1)
    u, v = p[:, 0] / p[:, 2], p[:, 1] / p[:, 2]
    r2 = u * u + v * v
    xd, yd = u, v
    if model in (CAM_PINHOLE_RADIAL1, CAM_PINHOLE_RADIAL3, CAM_PINHOLE_BROWN):
        c = 1 + intr[:, 3] * r2
        if model != CAM_PINHOLE_RADIAL1:
            c = c + intr[:, 4] * r2 * r2 + intr[:, 5] * r2 * r2 * r2
        xd, yd = u * c, v * c
        if model == CAM_PINHOLE_BROWN:
            t1, t2 = intr[:, 6], intr[:, 7]
            xd = xd + t2 * (r2 + 2 * u * u) + 2 * t1 * u * v
            yd = yd + t1 * (r2 + 2 * v * v) + 2 * t2 * u * v
    elif model == CAM_PINHOLE_FISHEYE:
        r = np.sqrt(r2)
        th = np.arctan(r)
        thd = th + intr[:, 3] * th ** 3 + intr[:, 4] * th ** 5 + intr[:, 5] * th ** 7 + intr[:, 6] * th ** 9
        cd = np.where(r > 1e-8, thd / np.where(r > 1e-8, r, 1.0), 1.0)
        xd, yd = u * cd, v * cd
    return np.stack([intr[:, 1] + intr[:, 0] * xd, intr[:, 2] + intr[:, 0] * yd], axis=1)


def geometric_track_lengths(n_points, mean=6.0, lo=2, hi=40, seed=0x7AC4):
    """track lengths of a scene with a realistic distribution: lo + geometric, mean `mean`, cut at `hi` (most tracks short, a long tail)"""
    rng = np.random.default_rng(seed)
    return np.minimum(lo + rng.geometric(1.0 / (mean - lo + 1.0), size=n_points) - 1, hi).astype(np.int64)


def ba_scene(n_cams, n_points, track_len=10, model=CAM_PINHOLE, n_intr_groups=1, seed=0xBA5E0000,
             noise_px=0.5, rot_deg=0.5, center_sigma=0.01, point_sigma=0.01, k_gt=(-0.05, 0.01, 0.0),
             n_rings=4, outlier_frac=0.0, track_lens=None):
    """Returns a dict with ground truth and the perturbed initial problem (flat arrays, mvgx_ba_problem layout).
    track_lens: per-point track lengths (array of n_points) instead of the one `track_len` of every point."""
    rng = np.random.default_rng(seed)
    if track_lens is not None:
        track_lens = np.minimum(np.asarray(track_lens, np.int64), n_cams)
        assert len(track_lens) == n_points and track_lens.min() >= 1
        track_len = int(track_lens.max())
    track_len = min(track_len, n_cams)
    n_rings = max(1, min(n_rings, n_cams // max(1, track_len)))
    # cameras: ring-major order, radii 1.5..3, small height offsets, looking at the origin
    per_ring = int(np.ceil(n_cams / n_rings))
    idx = np.arange(n_cams)
    ring, k = idx // per_ring, idx % per_ring
    radius = 1.5 + 1.5 * ring / max(1, n_rings - 1)
    ang = 2 * np.pi * (k + 0.25 * ring) / per_ring
    C = np.stack([radius * np.cos(ang), 0.3 * (ring - (n_rings - 1) / 2) + 0.01 * rng.standard_normal(n_cams),
                  radius * np.sin(ang)], axis=1)
    z = -C / np.linalg.norm(C, axis=1, keepdims=True)
    x = np.cross(np.array([0.0, 1.0, 0.0])[None], z); x /= np.linalg.norm(x, axis=1, keepdims=True)
    y = np.cross(z, x)
    Rgt = np.stack([x, y, z], axis=1)          # rows = camera axes: p_cam = R (X - C)
    aa_gt = _rotmat_to_aa(Rgt)
    t_gt = -np.einsum("nij,nj->ni", Rgt, C)
    poses_gt = np.concatenate([aa_gt, t_gt], axis=1)
    # intrinsics: groups of consecutive cameras share one intrinsic
    K = _NPARAM[model]
    intr_gt = np.zeros((n_intr_groups, 8))
    intr_gt[:, 0] = 1000.0; intr_gt[:, 1] = 500.0; intr_gt[:, 2] = 500.0
    if model == CAM_PINHOLE_RADIAL1:
        intr_gt[:, 3] = k_gt[0]
    if model in (CAM_PINHOLE_RADIAL3, CAM_PINHOLE_BROWN):
        intr_gt[:, 3:6] = k_gt
    if model == CAM_PINHOLE_BROWN:
        intr_gt[:, 6:8] = (0.002, -0.001)
    if model == CAM_PINHOLE_FISHEYE:
        intr_gt[:, 3:7] = (0.02, -0.004, 0.001, 0.0)
    if model == CAM_SPHERICAL:       # no parameter block: the row carries the image size {w, h}
        intr_gt[:, :] = 0.0
        intr_gt[:, 0] = 2000.0; intr_gt[:, 1] = 1000.0
    cam_group = (np.arange(n_cams) * n_intr_groups) // n_cams
    X_gt = rng.uniform(-0.3, 0.3, size=(n_points, 3))
    # visibility: `track_len` consecutive cameras (ring-major order, wrapping) from a random start
    start = rng.integers(0, n_cams, size=n_points)
    if track_lens is None:
        obs_point = np.repeat(np.arange(n_points, dtype=np.uint32), track_len)
        obs_pose = ((start[:, None] + np.arange(track_len)[None, :]) % n_cams).astype(np.uint32).reshape(-1)
    else:   # point p: track_lens[p] consecutive cameras from its start
        obs_point = np.repeat(np.arange(n_points, dtype=np.uint32), track_lens)
        first = np.concatenate([[0], np.cumsum(track_lens)[:-1]])
        within = np.arange(int(track_lens.sum()), dtype=np.int64) - np.repeat(first, track_lens)
        obs_pose = ((np.repeat(start, track_lens) + within) % n_cams).astype(np.uint32)
    obs_intr = cam_group[obs_pose].astype(np.uint32)
    xy = project(model, intr_gt[obs_intr], poses_gt[obs_pose], X_gt[obs_point])
    xy = xy + noise_px * rng.standard_normal(xy.shape)
    if outlier_frac > 0:
        bad = rng.random(len(xy)) < outlier_frac
        xy[bad] += 60.0 * rng.standard_normal((int(bad.sum()), 2))
    # perturbed initial state
    Rn = _rodrigues(np.deg2rad(rot_deg) * rng.standard_normal((n_cams, 3)))
    R0 = np.einsum("nij,njk->nik", Rn, Rgt)
    C0 = C + center_sigma * rng.standard_normal(C.shape)
    poses0 = np.concatenate([_rotmat_to_aa(R0), -np.einsum("nij,nj->ni", R0, C0)], axis=1)
    intr0 = intr_gt.copy()
    if model != CAM_SPHERICAL:
        intr0[:, 3:] = 0.0
    X0 = X_gt + point_sigma * rng.standard_normal(X_gt.shape)
    return {
        "n_poses": n_cams, "n_intrinsics": n_intr_groups, "n_points": n_points, "n_obs": len(obs_point),
        "poses": np.ascontiguousarray(poses0), "intrinsics": np.ascontiguousarray(intr0),
        "intr_model": np.full(n_intr_groups, model, np.int32), "points": np.ascontiguousarray(X0),
        "obs_pose": obs_pose, "obs_intr": obs_intr, "obs_point": obs_point, "obs_xy": np.ascontiguousarray(xy),
        "poses_gt": poses_gt, "intrinsics_gt": intr_gt, "points_gt": X_gt, "n_intr_params": K,
        "huber_a": 16.0,
    }


def add_control_points(scene, n_ctrl=6, views_per_point=4, weight=20.0, noise_px=0.0, seed=7):
    """Ground control points (SfM_Data::control_points + Control_Point_Parameter(weight, true), sfm_data_BA.hpp:44-64,
    sfm_data_BA_ceres.cpp:398-451): known 3-D positions, constant in the solve, observed in `views_per_point` images with
    weighted, loss-free residuals. Appended to the flat problem as extra points / observations. Returns a new dict."""
    rng = np.random.default_rng(seed)
    sc = dict(scene)
    n_poses, n_pts = int(scene["n_poses"]), int(scene["n_points"])
    Xc = rng.uniform(-0.3, 0.3, size=(n_ctrl, 3))
    cam_group = np.zeros(n_poses, np.uint32)
    cam_group[scene["obs_pose"]] = scene["obs_intr"]
    op = np.concatenate([rng.choice(n_poses, size=min(views_per_point, n_poses), replace=False) for _ in range(n_ctrl)]).astype(np.uint32)
    ox = np.repeat(np.arange(n_ctrl, dtype=np.uint32), min(views_per_point, n_poses))
    oi = cam_group[op]
    model = int(scene["intr_model"][0])
    xy = project(model, scene["intrinsics_gt"][oi], scene["poses_gt"][op], Xc[ox]) + noise_px * rng.standard_normal((len(op), 2))
    sc["points"] = np.concatenate([scene["points"], Xc]); sc["points_gt"] = np.concatenate([scene["points_gt"], Xc])
    sc["n_points"] = n_pts + n_ctrl
    sc["obs_pose"] = np.concatenate([scene["obs_pose"], op]); sc["obs_intr"] = np.concatenate([scene["obs_intr"], oi])
    sc["obs_point"] = np.concatenate([scene["obs_point"], ox + n_pts]).astype(np.uint32)
    sc["obs_xy"] = np.concatenate([scene["obs_xy"], xy]); sc["n_obs"] = len(sc["obs_pose"])
    n_old = int(scene["n_obs"])
    sc["obs_weight"] = np.concatenate([np.zeros(n_old), np.full(len(op), float(weight))])
    sc["obs_is_control"] = np.concatenate([np.zeros(n_old, np.uint8), np.ones(len(op), np.uint8)])
    sc["point_const_mask"] = np.concatenate([np.zeros(n_pts, np.uint8), np.ones(n_ctrl, np.uint8)])
    sc["n_structure_points"] = n_pts
    sc["control_weight"] = float(weight)
    return sc


def add_pose_priors(scene, weight=(1.0, 1.0, 1.0), sigma=0.0, huber_a=0.0, seed=11, every=1):
    """Pose-centre priors (ViewPriors::SetPoseCenterPrior, sfm_view_priors.hpp:55-63): prior centre = ground-truth centre
    (+ N(0, sigma)), one per `every`-th pose. `huber_a` = Square(pose_center_robust_fitting_error) of the reference."""
    rng = np.random.default_rng(seed)
    sc = dict(scene)
    Rgt = _rodrigues(scene["poses_gt"][:, :3])
    Cgt = -np.einsum("nji,nj->ni", Rgt, scene["poses_gt"][:, 3:6])
    idx = np.arange(0, int(scene["n_poses"]), every, dtype=np.uint32)
    sc["prior_pose"] = idx
    sc["prior_center"] = np.ascontiguousarray(Cgt[idx] + sigma * rng.standard_normal((len(idx), 3)))
    sc["prior_weight"] = np.tile(np.asarray(weight, np.float64), (len(idx), 1))
    sc["prior_huber_a"] = float(huber_a)
    return sc


def binary_descriptors(n_images, n_desc=300, n_bytes=64, seed=0, n_world=None, flip_bits=40):
    """AKAZE-MLDB-like packed binary descriptors: every image sees a window of shared 'world' bit strings with `flip_bits`
    random bit flips (true correspondences: small Hamming distance) plus unrelated rows. n_desc: int or per-image list."""
    rng = np.random.default_rng(seed)
    sizes = [int(n_desc)] * n_images if np.isscalar(n_desc) else [int(v) for v in n_desc]
    n_world = n_world or max(4 * max(sizes + [1]), 16)
    world = rng.integers(0, 256, (n_world, n_bytes), dtype=np.uint8)
    out = []
    for k, n in enumerate(sizes):
        if n == 0:
            out.append(np.zeros((0, n_bytes), np.uint8))
            continue
        idx = (rng.integers(0, max(n_world // 2, 1)) + rng.permutation(max(n_world // 2, 1))[:n]) % n_world
        if len(idx) < n:
            idx = np.concatenate([idx, rng.integers(0, n_world, n - len(idx))])
        d = world[idx].copy()
        bits = rng.integers(0, n_bytes * 8, (n, flip_bits))
        for j in range(flip_bits):
            d[np.arange(n), bits[:, j] >> 3] ^= (1 << (bits[:, j] & 7)).astype(np.uint8)
        fresh = rng.random(n) < 0.3
        d[fresh] = rng.integers(0, 256, (int(fresh.sum()), n_bytes), dtype=np.uint8)
        out.append(np.ascontiguousarray(d))
    return out


def float_descriptors(n_images, n_desc=300, dim=64, seed=0, noise=0.05):
    """AKAZE-MSURF-like float descriptors: unit-norm rows drawn from a shared set of 'world' vectors plus noise (true
    correspondences) and unrelated rows. n_desc: int or per-image list. float32, C-contiguous."""
    rng = np.random.default_rng(seed)
    sizes = [int(n_desc)] * n_images if np.isscalar(n_desc) else [int(v) for v in n_desc]
    n_world = max(4 * max(sizes + [1]), 16)
    world = rng.standard_normal((n_world, dim)).astype(np.float32)
    out = []
    for n in sizes:
        if n == 0:
            out.append(np.zeros((0, dim), np.float32))
            continue
        idx = rng.permutation(n_world // 2)[:n]
        if len(idx) < n:
            idx = np.concatenate([idx, rng.integers(0, n_world, n - len(idx))])
        d = world[idx] + noise * rng.standard_normal((n, dim)).astype(np.float32)
        fresh = rng.random(n) < 0.3
        d[fresh] = rng.standard_normal((int(fresh.sum()), dim)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        out.append(np.ascontiguousarray(d.astype(np.float32)))
    return out


# ---------------------------------------------------------------------------------------------------------
# Two-view correspondences for the geometric filter: per image pair two pinhole views of a random point cloud
# (rotation about y, baseline along x), pixel noise on the second view, a share of uniformly random outliers; some pairs have
# no geometry at all (the filter's early exit), some fewer than 8 correspondences (rejected without estimation).
# ---------------------------------------------------------------------------------------------------------
def two_view_matches(n_pairs, seed=0, n_min=8, n_max=400, inlier_frac=(0.3, 0.9), noise_px=0.4, no_geometry_frac=0.25, tiny_frac=0.02,
                     sizes=((1000, 1000), (1280, 960), (1920, 1080))):
    """Returns dict(xI, xJ: (N, 2) float64 pixels; start: (n_pairs + 1,) uint64; wh: (n_pairs, 4) uint32 {w_I, h_I, w_J, h_J};
    is_inlier: (N,) bool ground truth)."""
    rng = np.random.default_rng(seed)
    xs_i, xs_j, truth, start, wh = [], [], [], [0], []
    for _ in range(n_pairs):
        u = rng.random()
        n = int(rng.integers(0, 8)) if u < tiny_frac else int(rng.integers(n_min, n_max + 1))
        (wi, hi), (wj, hj) = sizes[rng.integers(len(sizes))], sizes[rng.integers(len(sizes))]
        f = 0.9 * max(wi, hi)
        th = rng.uniform(0.03, 0.25) * (1 if rng.random() < 0.5 else -1)
        base = rng.uniform(0.3, 1.0)
        X = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), 6 + rng.uniform(-1.5, 1.5, n)], 1)
        Y = np.stack([np.cos(th) * X[:, 0] + np.sin(th) * X[:, 2] - base, X[:, 1] + 0.05, -np.sin(th) * X[:, 0] + np.cos(th) * X[:, 2]], 1)
        a = np.stack([f * X[:, 0] / X[:, 2] + wi / 2, f * X[:, 1] / X[:, 2] + hi / 2], 1)
        b = np.stack([f * Y[:, 0] / Y[:, 2] + wj / 2, f * Y[:, 1] / Y[:, 2] + hj / 2], 1) + rng.normal(0, noise_px, (n, 2))
        frac = 0.0 if rng.random() < no_geometry_frac else rng.uniform(*inlier_frac)
        inl = rng.random(n) < frac
        b[~inl] = np.stack([rng.uniform(0, wj, (~inl).sum()), rng.uniform(0, hj, (~inl).sum())], 1)
        xs_i.append(a); xs_j.append(b); truth.append(inl)
        start.append(start[-1] + n)
        wh.append((wi, hi, wj, hj))
    cat = lambda v, w: np.ascontiguousarray(np.concatenate(v) if v else np.zeros((0, w)), dtype=np.float64)   # noqa: E731
    return dict(xI=cat(xs_i, 2).reshape(-1, 2), xJ=cat(xs_j, 2).reshape(-1, 2), start=np.asarray(start, np.uint64),
                wh=np.asarray(wh, np.uint32).reshape(-1, 4), is_inlier=np.concatenate(truth) if truth else np.zeros(0, bool))


def two_view_calibration(tv):
    """The calibration matrices of the cameras two_view_matches / two_view_matches_bulk projected with: f = 0.9 max(w_I, h_I) for both
    images of a pair, principal point at the image centre. Returns (n_pairs, 2, 3, 3) float64 {K_I, K_J}."""
    wh = np.asarray(tv["wh"], np.float64).reshape(-1, 4)
    f = 0.9 * np.maximum(wh[:, 0], wh[:, 1])
    K = np.zeros((len(wh), 2, 3, 3))
    K[:, :, 0, 0] = f[:, None]; K[:, :, 1, 1] = f[:, None]; K[:, :, 2, 2] = 1.0
    K[:, 0, 0, 2] = wh[:, 0] / 2; K[:, 0, 1, 2] = wh[:, 1] / 2; K[:, 1, 0, 2] = wh[:, 2] / 2; K[:, 1, 1, 2] = wh[:, 3] / 2
    return K


def two_view_homography_matches(n_pairs, seed=0, n_min=5, n_max=400, inlier_frac=(0.3, 0.9), noise_px=0.4, no_geometry_frac=0.25, tiny_frac=0.03,
                                sizes=((1000, 1000), (1280, 960), (1920, 1080))):
    """Putative matches of image pairs related by a homography (a plane seen from two views / a rotating camera): x_J ~ H x_I + noise
    for the inliers, uniform positions for the rest. Same dict as two_view_matches."""
    rng = np.random.default_rng(seed)
    xs_i, xs_j, truth, start, wh = [], [], [], [0], []
    for _ in range(n_pairs):
        u = rng.random()
        n = int(rng.integers(0, 5)) if u < tiny_frac else int(rng.integers(n_min, n_max + 1))
        (wi, hi), (wj, hj) = sizes[rng.integers(len(sizes))], sizes[rng.integers(len(sizes))]
        th = rng.uniform(-0.3, 0.3); sc = rng.uniform(0.8, 1.25)
        H = np.array([[sc * np.cos(th), -sc * np.sin(th), rng.uniform(-0.1, 0.1) * wj],
                      [sc * np.sin(th), sc * np.cos(th), rng.uniform(-0.1, 0.1) * hj],
                      [rng.uniform(-1e-4, 1e-4), rng.uniform(-1e-4, 1e-4), 1.0]])
        a = np.stack([rng.uniform(0, wi, n), rng.uniform(0, hi, n)], 1)
        ah = np.concatenate([a, np.ones((n, 1))], 1) @ H.T
        b = ah[:, :2] / ah[:, 2:3] + rng.normal(0, noise_px, (n, 2))
        frac = 0.0 if rng.random() < no_geometry_frac else rng.uniform(*inlier_frac)
        inl = rng.random(n) < frac
        b[~inl] = np.stack([rng.uniform(0, wj, (~inl).sum()), rng.uniform(0, hj, (~inl).sum())], 1)
        xs_i.append(a); xs_j.append(b); truth.append(inl)
        start.append(start[-1] + n)
        wh.append((wi, hi, wj, hj))
    cat = lambda v, w: np.ascontiguousarray(np.concatenate(v) if v else np.zeros((0, w)), dtype=np.float64)   # noqa: E731
    return dict(xI=cat(xs_i, 2).reshape(-1, 2), xJ=cat(xs_j, 2).reshape(-1, 2), start=np.asarray(start, np.uint64),
                wh=np.asarray(wh, np.uint32).reshape(-1, 4), is_inlier=np.concatenate(truth) if truth else np.zeros(0, bool))


def two_view_matches_bulk(n_pairs, n=250, seed=0, inlier_frac=(0.3, 0.9), noise_px=0.4, no_geometry_frac=0.25, size=(1000, 1000)):
    """The same kind of scene as two_view_matches with n correspondences in every pair, generated in bulk (bench workloads)."""
    rng = np.random.default_rng(seed)
    w, h = size
    f = 0.9 * max(w, h)
    th = rng.uniform(0.03, 0.25, (n_pairs, 1)) * np.where(rng.random((n_pairs, 1)) < 0.5, 1.0, -1.0)
    base = rng.uniform(0.3, 1.0, (n_pairs, 1))
    X0, X1, X2 = rng.uniform(-2, 2, (n_pairs, n)), rng.uniform(-1.5, 1.5, (n_pairs, n)), 6 + rng.uniform(-1.5, 1.5, (n_pairs, n))
    Y0, Y1, Y2 = np.cos(th) * X0 + np.sin(th) * X2 - base, X1 + 0.05, -np.sin(th) * X0 + np.cos(th) * X2
    a = np.stack([f * X0 / X2 + w / 2, f * X1 / X2 + h / 2], 2)
    b = np.stack([f * Y0 / Y2 + w / 2, f * Y1 / Y2 + h / 2], 2) + rng.normal(0, noise_px, (n_pairs, n, 2))
    frac = np.where(rng.random((n_pairs, 1)) < no_geometry_frac, 0.0, rng.uniform(*inlier_frac, (n_pairs, 1)))
    inl = rng.random((n_pairs, n)) < frac
    out = np.stack([rng.uniform(0, w, (n_pairs, n)), rng.uniform(0, h, (n_pairs, n))], 2)
    b = np.where(inl[:, :, None], b, out)
    return dict(xI=np.ascontiguousarray(a.reshape(-1, 2)), xJ=np.ascontiguousarray(b.reshape(-1, 2)),
                start=(np.arange(n_pairs + 1, dtype=np.uint64) * np.uint64(n)), wh=np.tile(np.array([w, h, w, h], np.uint32), (n_pairs, 1)),
                is_inlier=inl.reshape(-1))


def sfm_image_set(n_images, n_desc=2000, visible=1200, stride=120, seed=0x5F3D5E7, size=(2000, 1500), noise_px=0.5, desc_noise=3):
    """A consistent image collection for the matching -> geometric-filter pipeline: cameras on an arc looking at a cloud of
    landmarks; image k sees the landmarks [k * stride, k * stride + visible) (neighbours share most of them, images more than
    visible / stride apart share none) plus n_desc - visible clutter features. A landmark's descriptor is its world descriptor
    + integer noise, its position the pinhole projection + Gaussian noise; clutter has random descriptors and positions.
    Returns dict(desc=[(n_desc, 128) u8], xy=[(n_desc, 2) f64], landmark=[(n_desc,) int64, -1 for clutter], size=(w, h))."""
    rng = np.random.default_rng(seed)
    w, h = size
    f = 0.9 * max(w, h)
    n_land = (n_images - 1) * stride + visible
    world = world_descriptors(n_land, seed=seed ^ 0x1111)
    ang = rng.uniform(0, 2 * np.pi, n_land)   # landmarks in a ball of radius 1.5 around the origin
    X = np.stack([rng.uniform(-1.5, 1.5, n_land), rng.uniform(-1.0, 1.0, n_land), rng.uniform(-1.5, 1.5, n_land)], 1)
    del ang
    desc, xy, land = [], [], []
    for k in range(n_images):
        r = np.random.default_rng(seed + 1 + k)
        th = 0.004 * k   # the cameras move along an arc of radius 8, looking at the origin
        c = np.array([8.0 * np.sin(th), 0.3 * np.sin(0.05 * k), -8.0 * np.cos(th)])
        zc = -c / np.linalg.norm(c); xc = np.cross([0.0, 1.0, 0.0], zc); xc /= np.linalg.norm(xc); yc = np.cross(zc, xc)
        R = np.stack([xc, yc, zc])
        ids = np.arange(k * stride, k * stride + visible)
        P = (X[ids] - c) @ R.T
        uv = np.stack([f * P[:, 0] / P[:, 2] + w / 2, f * P[:, 1] / P[:, 2] + h / 2], 1) + r.normal(0, noise_px, (visible, 2))
        d = np.clip(world[ids].astype(np.int16) + r.integers(-desc_noise, desc_noise + 1, (visible, 128), dtype=np.int16), 0, 255).astype(np.uint8)
        n_cl = n_desc - visible
        dc = world_descriptors(n_cl, seed=(seed ^ 0x2222) + k)
        uvc = np.stack([r.uniform(0, w, n_cl), r.uniform(0, h, n_cl)], 1)
        perm = r.permutation(n_desc)
        desc.append(np.ascontiguousarray(np.concatenate([d, dc])[perm]))
        xy.append(np.ascontiguousarray(np.concatenate([uv, uvc])[perm]))
        land.append(np.concatenate([ids, -np.ones(n_cl, np.int64)])[perm])
    return dict(desc=desc, xy=xy, landmark=land, size=(w, h))
