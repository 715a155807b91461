"""openMVG's BA option enums and their translation to the constant-component masks of mvgx_ba_problem.

  Extrinsic_Parameter_Type / Structure_Parameter_Type / Optimize_Options   sfm/sfm_data_BA.hpp:20-89
  cameras::Intrinsic_Parameter_Type                                          cameras/Camera_Common.hpp:92-100
  IntrinsicBase::subsetParameterization                                      cameras/Camera_Pinhole.hpp:244-261,
                                                                             cameras/Camera_Pinhole_Radial.hpp (K1, K3 :405-428)
  pose block subset handling                                                 sfm/sfm_data_BA_ceres.cpp:274-306
"""
from enum import IntFlag

import numpy as np


class Intrinsic_Parameter_Type(IntFlag):
    NONE = 1
    ADJUST_FOCAL_LENGTH = 2
    ADJUST_PRINCIPAL_POINT = 4
    ADJUST_DISTORTION = 8
    ADJUST_ALL = 14


class Extrinsic_Parameter_Type(IntFlag):
    NONE = 1
    ADJUST_ROTATION = 2
    ADJUST_TRANSLATION = 4
    ADJUST_ALL = 6


class Structure_Parameter_Type(IntFlag):
    NONE = 0
    ADJUST_ALL = 1


# PINHOLE_CAMERA, _RADIAL1, _RADIAL3, _BROWN, _FISHEYE, CAMERA_SPHERICAL (no parameter block)
N_INTR_PARAMS = {1: 3, 2: 4, 3: 6, 4: 8, 5: 7, 7: 0}


def pose_const_mask(extrinsics_opt):
    e = int(extrinsics_opt)
    if e == Extrinsic_Parameter_Type.NONE:
        return 0x3F                      # SetParameterBlockConstant
    if e == Extrinsic_Parameter_Type.ADJUST_TRANSLATION:
        return 0x07                      # rotation {0,1,2} constant
    if e == Extrinsic_Parameter_Type.ADJUST_ROTATION:
        return 0x38                      # translation {3,4,5} constant
    return 0


def intr_const_mask(model, intrinsics_opt):
    p = int(intrinsics_opt)
    K = N_INTR_PARAMS[int(model)]
    if p == Intrinsic_Parameter_Type.NONE:
        return (1 << K) - 1              # SetParameterBlockConstant (sfm_data_BA_ceres.cpp:321-325)
    m = 0
    if not (p & Intrinsic_Parameter_Type.ADJUST_FOCAL_LENGTH) or (p & Intrinsic_Parameter_Type.NONE):
        m |= 0x1
    if not (p & Intrinsic_Parameter_Type.ADJUST_PRINCIPAL_POINT) or (p & Intrinsic_Parameter_Type.NONE):
        m |= 0x6
    if K > 3 and (not (p & Intrinsic_Parameter_Type.ADJUST_DISTORTION) or (p & Intrinsic_Parameter_Type.NONE)):
        m |= ((1 << K) - 1) & ~0x7
    return m


def masks_for(scene, intrinsics_opt=Intrinsic_Parameter_Type.ADJUST_ALL, extrinsics_opt=Extrinsic_Parameter_Type.ADJUST_ALL,
              structure_opt=Structure_Parameter_Type.ADJUST_ALL):
    pm = np.full(int(scene["n_poses"]), pose_const_mask(extrinsics_opt), np.uint8)
    im = np.array([intr_const_mask(m, intrinsics_opt) for m in scene["intr_model"]], np.uint8)
    return {"pose_const_mask": pm, "intr_const_mask": im, "points_constant": int(structure_opt) == 0}


def _rot(aa):
    from .synth import _rodrigues
    return _rodrigues(aa)


def writeback_poses(poses_before, poses_solved, extrinsics_opt):
    """Pose write-back of Bundle_Adjustment_Ceres::Adjust (sfm_data_BA_ceres.cpp:527-556) on [angle-axis, t] blocks.

    The solver optimises [aa, t] with t = -R C. The reference then updates its Pose3(R, C):
      ADJUST_ALL         : R <- R_refined, C <- -R_refined^T t_refined
      ADJUST_TRANSLATION : C <- -R_refined^T t_refined           (rotation block was constant)
      ADJUST_ROTATION    : R <- R_refined and the old CENTRE is kept (so t becomes -R_refined C_old, not the solved t)
      NONE               : nothing
    Returns the [aa, t] blocks of the scene after write-back."""
    e = int(extrinsics_opt)
    before = np.asarray(poses_before, np.float64)
    solved = np.asarray(poses_solved, np.float64)
    if e == Extrinsic_Parameter_Type.NONE:
        return before.copy()
    if e == Extrinsic_Parameter_Type.ADJUST_ROTATION:
        R_old = _rot(before[:, :3])
        C_old = -np.einsum("nji,nj->ni", R_old, before[:, 3:6])
        R_new = _rot(solved[:, :3])
        out = solved.copy()
        out[:, 3:6] = -np.einsum("nij,nj->ni", R_new, C_old)
        return out
    return solved.copy()
