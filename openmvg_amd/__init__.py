"""openmvg_amd — MI355X (gfx950) accelerators behind two openMVG interfaces.

  matching : brute-force L2 2-NN + Lowe ratio on 128-D uint8 descriptors  (openmvg_amd.matching)
  ba       : Levenberg-Marquardt bundle adjustment                         (openmvg_amd.ba)

All compute lives in openmvg_amd/lib/libmvgx_hip.so (hand-written HIP, C ABI in include/mvgx.h);
the Python modules only mirror the reference's host interfaces. No CPU fallback exists.
"""
__version__ = "0.1.0"
