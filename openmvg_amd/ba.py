"""Host-side mirror of openMVG's bundle-adjustment interface on top of the mvgx C ABI.

  Bundle_Adjustment (abstract: bool Adjust(SfM_Data&, const Optimize_Options&))   sfm/sfm_data_BA.hpp:92-105
  Bundle_Adjustment_Ceres / BA_Ceres_options                                       sfm/sfm_data_BA_ceres.hpp:31-69
  Optimize_Options                                                                  sfm/sfm_data_BA.hpp:66-89

The scene is the flat form of SfM_Data the C ABI takes (dict: poses [aa, t], intrinsics, intr_model, points,
obs_pose / obs_intr / obs_point / obs_xy — layouts of sfm_data_BA_ceres.cpp:260-396). All numerics run in
libmvgx_hip.so on the GPU; there is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _capi
from . import ba_options as bo


class Optimize_Options:
    def __init__(self, intrinsics_opt=bo.Intrinsic_Parameter_Type.ADJUST_ALL, extrinsics_opt=bo.Extrinsic_Parameter_Type.ADJUST_ALL,
                 structure_opt=bo.Structure_Parameter_Type.ADJUST_ALL):
        self.intrinsics_opt = intrinsics_opt
        self.extrinsics_opt = extrinsics_opt
        self.structure_opt = structure_opt


class BA_Ceres_options:
    """Field names of Bundle_Adjustment_Ceres::BA_Ceres_options (sfm_data_BA_ceres.cpp:110-149)."""

    def __init__(self, bVerbose=False, bmultithreaded=True):
        self.bVerbose_ = bVerbose
        self.parameter_tolerance_ = 1e-8
        self.gradient_tolerance_ = 1e-10
        self.bUse_loss_function_ = True
        self.max_num_iterations_ = 50


def default_options(**kw):
    o = _capi.BaOptions()
    _capi.lib().mvgx_ba_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def comm_unique_id():
    """128-byte ncclUniqueId (call on rank 0, broadcast to the other ranks)."""
    buf = (C.c_char * 128)()
    _capi.check(_capi.lib().mvgx_comm_unique_id(C.cast(buf, C.c_void_p)))
    return bytes(buf.raw)


class BaContext:
    """Device-resident BA problem (thin wrapper over mvgx_ba_*)."""

    def __init__(self, scene, pose_const_mask=None, intr_const_mask=None, points_constant=False, huber_a=16.0, device=-1,
                 devices=None):
        """device: one ordinal (-1: MVGX_DEVICES or the current device). devices: list of ordinals -> one context over several
        devices of this process (mvgx_ba_create_multi: the problem is sharded inside the library)."""
        p = self._problem(scene, pose_const_mask, intr_const_mask, points_constant, huber_a)
        self.shape = (p.n_poses, p.n_intrinsics, p.n_points)
        self._h = C.c_void_p()
        if devices is not None:
            arr_d = (C.c_int * len(devices))(*[int(x) for x in devices])
            _capi.check(_capi.lib().mvgx_ba_create_multi(arr_d, len(devices), C.byref(p), C.byref(self._h)))
        else:
            _capi.check(_capi.lib().mvgx_ba_create(int(device), C.byref(p), C.byref(self._h)))

    def _problem(self, scene, pose_const_mask, intr_const_mask, points_constant, huber_a):
        """mvgx_ba_problem over the scene's arrays (kept alive in self._keep until the next create / update)"""
        keep = {}

        def arr(name, dtype):
            a = np.ascontiguousarray(scene[name], dtype=dtype)
            keep[name] = a
            return a.ctypes.data

        p = _capi.BaProblem()
        p.n_poses = int(scene["n_poses"]); p.n_intrinsics = int(scene["n_intrinsics"]); p.n_points = int(scene["n_points"])
        p.n_obs = int(scene["n_obs"])
        p.poses = arr("poses", np.float64); p.intrinsics = arr("intrinsics", np.float64)
        p.intr_model = arr("intr_model", np.int32); p.points = arr("points", np.float64)
        p.obs_pose = arr("obs_pose", np.uint32); p.obs_intr = arr("obs_intr", np.uint32); p.obs_point = arr("obs_point", np.uint32)
        p.obs_xy = arr("obs_xy", np.float64)
        if pose_const_mask is not None:
            keep["pm"] = np.ascontiguousarray(pose_const_mask, np.uint8); p.pose_const_mask = keep["pm"].ctypes.data
        if intr_const_mask is not None:
            keep["im"] = np.ascontiguousarray(intr_const_mask, np.uint8); p.intr_const_mask = keep["im"].ctypes.data
        p.points_constant = 1 if points_constant else 0
        p.huber_a = float(huber_a)
        # optional: ground control points (weighted, loss-free residuals on constant points) and pose-centre priors
        if scene.get("obs_weight") is not None:
            p.obs_weight = arr("obs_weight", np.float64)
        if scene.get("obs_is_control") is not None:
            p.obs_is_control = arr("obs_is_control", np.uint8)
        if scene.get("point_const_mask") is not None:
            p.point_const_mask = arr("point_const_mask", np.uint8)
        if scene.get("prior_pose") is not None and len(scene["prior_pose"]):
            p.n_pose_priors = len(scene["prior_pose"])
            p.prior_pose = arr("prior_pose", np.uint32)
            p.prior_center = arr("prior_center", np.float64)
            p.prior_weight = arr("prior_weight", np.float64)
            p.prior_huber_a = float(scene.get("prior_huber_a", 0.0))
        self._keep = keep
        return p

    def update(self, scene, pose_const_mask=None, intr_const_mask=None, points_constant=False, huber_a=16.0, obs_enabled=None):
        """mvgx_ba_update: new values (parameters, image points, weights, prior targets, constant masks, loss scale) for the structure
        this context was created from. Returns False - and leaves the context as it was - when the scene's structure differs
        (MVGX_ERR_STRUCTURE): the caller closes this context and creates a new one."""
        keep_before = self._keep
        p = self._problem(scene, pose_const_mask, intr_const_mask, points_constant, huber_a)
        if obs_enabled is not None:   # mvgx_ba_update_subset: observations switched off without changing the structure
            en = np.ascontiguousarray(obs_enabled, np.uint8)
            assert en.shape == (int(scene["n_obs"]),)
            self._keep["obs_enabled"] = en
            rc = _capi.lib().mvgx_ba_update_subset(self._h, C.byref(p), en.ctypes.data)
        else:
            rc = _capi.lib().mvgx_ba_update(self._h, C.byref(p))
        if rc == _capi.MVGX_ERR_STRUCTURE:
            self._keep = keep_before
            return False
        _capi.check(rc)
        return True

    def comm_init(self, world, rank, unique_id):
        """Bind this rank's context to an RCCL communicator (unique_id: the 128 bytes of comm_unique_id() of rank 0)."""
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        _capi.check(_capi.lib().mvgx_ba_comm_init(self._h, int(world), int(rank), C.cast(buf, C.c_void_p)))

    def set_allreduce(self, fn):
        """Callback transport: fn(device_ptr, count, op, hip_stream) -> 0 on success (op: 0 sum, 1 max)."""
        self._cb = _capi.ALLREDUCE_F64(lambda _u, ptr, count, op, stream: int(fn(ptr, int(count), int(op), stream)))
        _capi.check(_capi.lib().mvgx_ba_set_allreduce(self._h, self._cb, None))

    def close(self):
        if self._h:
            _capi.lib().mvgx_ba_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solve(self, options=None):
        s = _capi.BaSummary()
        _capi.check(_capi.lib().mvgx_ba_solve(self._h, C.byref(options or default_options()), C.byref(s)))
        return s

    def lm_iteration(self, options=None):
        s = _capi.BaSummary()
        _capi.check(_capi.lib().mvgx_ba_lm_iteration(self._h, C.byref(options or default_options()), C.byref(s)))
        return s

    def evaluate(self):
        cost, rmse = C.c_double(), C.c_double()
        _capi.check(_capi.lib().mvgx_ba_evaluate(self._h, C.byref(cost), C.byref(rmse)))
        return cost.value, rmse.value

    def residuals(self):
        """per-observation pixel residual norms at the current parameters (order of the scene's observation arrays)"""
        out = np.zeros(int(self._keep["obs_pose"].shape[0]))
        _capi.check(_capi.lib().mvgx_ba_residuals(self._h, out.ctypes.data))
        return out

    def track_angles(self):
        """per point: largest angle (degrees) between the world rays of two of its observations (mvgx_ba_track_angles)"""
        out = np.zeros(int(self.shape[2]))
        _capi.check(_capi.lib().mvgx_ba_track_angles(self._h, out.ctypes.data))
        return out

    LINEAR_SOLVERS = {"auto": 0, "dense": 1, "sparse": 2, "sparse_preferred": 3}

    def set_linear_solver(self, kind):
        """the caller's linear_solver_type_ (sfm_data_BA_ceres.cpp:132-146,483): "auto", "dense" (DENSE_SCHUR), "sparse" /
        "sparse_preferred" (SPARSE_SCHUR) - mvgx_ba_set_linear_solver; before the first iteration, or naming the solver in place"""
        _capi.check(_capi.lib().mvgx_ba_set_linear_solver(self._h, self.LINEAR_SOLVERS.get(kind, kind)))

    def solver_info(self):
        """how the reduced camera system is solved (mvgx_ba_get_solver_info; valid after the first iteration)"""
        info = _capi.BaSolverInfo()
        _capi.check(_capi.lib().mvgx_ba_get_solver_info(self._h, C.byref(info)))
        return info

    def read_params(self):
        npz, ni, nx = self.shape
        poses = np.zeros((npz, 6)); intr = np.zeros((ni, 8)); pts = np.zeros((nx, 3))
        _capi.check(_capi.lib().mvgx_ba_read_params(self._h, poses.ctypes.data, intr.ctypes.data, pts.ctypes.data))
        return poses, intr, pts


class Bundle_Adjustment_HIP:
    """Drop-in mirror of sfm::Bundle_Adjustment_Ceres: Adjust(scene, Optimize_Options) -> bool, scene updated in place
    with the reference's write-back rules (sfm_data_BA_ceres.cpp:527-568)."""

    def __init__(self, options=None, device=-1):
        self.ceres_options_ = options or BA_Ceres_options()
        self._device = device
        self.summary = None

    def ceres_options(self):
        return self.ceres_options_

    def Adjust(self, scene, options=None):
        options = options or Optimize_Options()
        # sfm_data_BA_ceres.cpp:84-108,388-392: a camera model without a cost functor makes Adjust() return false
        if any(int(m) not in bo.N_INTR_PARAMS for m in scene["intr_model"]):
            return False
        masks = bo.masks_for(scene, options.intrinsics_opt, options.extrinsics_opt, options.structure_opt)
        co = self.ceres_options_
        try:
            ctx = BaContext(scene, huber_a=16.0 if co.bUse_loss_function_ else 0.0, device=self._device, **masks)
        except _capi.MvgxError as e:
            if e.code == _capi.MVGX_ERR_UNSUPPORTED:   # "Cannot create a CostFunction for this camera model" -> false
                return False
            raise
        try:
            opt = default_options(max_num_iterations=co.max_num_iterations_, parameter_tolerance=co.parameter_tolerance_,
                                  gradient_tolerance=co.gradient_tolerance_)
            try:
                self.summary = ctx.solve(opt)
            except _capi.MvgxError as e:
                if e.code == _capi.MVGX_ERR_NUMERIC:   # !summary.IsSolutionUsable() -> false, poses/intrinsics untouched
                    return False
                raise
            poses, intr, pts = ctx.read_params()
        finally:
            ctx.close()
        before = np.array(scene["poses"], np.float64)
        if int(options.extrinsics_opt) != int(bo.Extrinsic_Parameter_Type.NONE):
            scene["poses"] = bo.writeback_poses(before, poses, options.extrinsics_opt)
        if int(options.intrinsics_opt) != int(bo.Intrinsic_Parameter_Type.NONE):
            scene["intrinsics"] = intr
        scene["points"] = pts
        return True


def RemoveOutliers_PixelResidualError(scene, dThresholdPixel, minTrackLength=2, device=-1):
    """Mirror of sfm::RemoveOutliers_PixelResidualError (sfm/sfm_data_filters.cpp:40-73) on the flat scene: observations
    whose reprojection residual norm exceeds the threshold are erased, then tracks with fewer than minTrackLength
    observations. The residuals come from the device (mvgx_ba_residuals). Returns (outlier_count, filtered scene); point
    ids are kept (erased tracks simply lose all their observations), as the reference keeps landmark ids."""
    ctx = BaContext(scene, device=device)
    try:
        res = ctx.residuals()
    finally:
        ctx.close()
    keep = ~(res > float(dThresholdPixel))
    outlier_count = int((~keep).sum())
    cnt = np.bincount(np.asarray(scene["obs_point"])[keep], minlength=int(scene["n_points"]))
    keep &= cnt[np.asarray(scene["obs_point"])] >= max(int(minTrackLength), 1)
    out = dict(scene)
    for k in ("obs_pose", "obs_intr", "obs_point", "obs_weight", "obs_is_control"):
        if scene.get(k) is not None:
            out[k] = np.ascontiguousarray(np.asarray(scene[k])[keep])
    out["obs_xy"] = np.ascontiguousarray(np.asarray(scene["obs_xy"], np.float64).reshape(-1, 2)[keep])
    out["n_obs"] = int(keep.sum())
    return outlier_count, out


def _drop_observations(scene, keep):
    out = dict(scene)
    for k in ("obs_pose", "obs_intr", "obs_point", "obs_weight", "obs_is_control"):
        if scene.get(k) is not None:
            out[k] = np.ascontiguousarray(np.asarray(scene[k])[keep])
    out["obs_xy"] = np.ascontiguousarray(np.asarray(scene["obs_xy"], np.float64).reshape(-1, 2)[keep])
    out["n_obs"] = int(keep.sum())
    return out


def RemoveOutliers_AngleError(scene, dMinAcceptedAngle, device=-1):
    """Mirror of sfm::RemoveOutliers_AngleError (sfm/sfm_data_filters.cpp:77-121) on the flat scene: tracks whose largest
    pairwise ray angle (device: mvgx_ba_track_angles) is below dMinAcceptedAngle degrees are erased - they lose all their
    observations, the point id stays. Tracks that are already empty do not exist in the reference's landmark map and are
    not counted. Returns (removed_track_count, filtered scene)."""
    ctx = BaContext(scene, device=device)
    try:
        ang = ctx.track_angles()
    finally:
        ctx.close()
    obs_point = np.asarray(scene["obs_point"])
    alive = np.bincount(obs_point, minlength=int(scene["n_points"])) > 0
    bad = alive & (ang < float(dMinAcceptedAngle))
    return int(bad.sum()), _drop_observations(scene, ~bad[obs_point])


def badTrackRejector(scene, dPrecision, count=0, device=-1):
    """SequentialSfMReconstructionEngine::badTrackRejector (sfm/pipelines/sequential/sequential_SfM.cpp:1226-1232):
    RemoveOutliers_PixelResidualError(dPrecision, 2) then RemoveOutliers_AngleError(2.0); True when more than `count`
    outliers were removed, i.e. when the caller's `do { BA } while (badTrackRejector(...))` loop runs again
    (:206-210). Returns (again, filtered scene)."""
    n_res, scene = RemoveOutliers_PixelResidualError(scene, dPrecision, 2, device=device)
    n_ang, scene = RemoveOutliers_AngleError(scene, 2.0, device=device)
    return (n_res + n_ang) > int(count), scene
