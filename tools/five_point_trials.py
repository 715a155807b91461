"""Five-point solver of the device (mvgx_debug_five_point: one wave, one sample) against the reference's FivePointsRelativePose
(oracle/_ref/libref_geofilter.so::ref_five_point) on random samples: clean two-view geometry and, with a third argument, every
second sample contaminated with 1-4 random bearings (what AC-RANSAC draws most of the time). Counts samples whose number of
solutions differs or whose models differ by more than 1e-8, and the solves whose eigenvalues fell back to hqr.
Usage: five_point_trials.py <seed> <n> [c]   (MVGX_TRIALS_EMU=1: the HIP emulation instead of the GPU)"""
import os, sys, time, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ref=C.CDLL(os.path.join(ROOT, 'oracle', '_ref', 'libref_geofilter.so'))
if os.environ.get("MVGX_TRIALS_EMU") == "1":
    from tests import _emu
    h=_emu.handle()
else:
    from openmvg_amd import _capi
    h=_capi.lib()
def P(a): return a.ctypes.data_as(C.c_void_p)
def norm(E):
    E=E/np.linalg.norm(E); k=np.abs(E).argmax(); return E*np.sign(E.flat[k])
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
worst=0; nbad=0
for trial in range(int(sys.argv[2]) if len(sys.argv)>2 else 20):
    # random relative pose, 5 points in front of both cameras
    R,_=np.linalg.qr(rng.normal(size=(3,3))); R*=np.sign(np.linalg.det(R))
    ang=rng.uniform(0.05,0.5); 
    from scipy.spatial.transform import Rotation
    R=Rotation.from_rotvec(rng.normal(size=3)*ang).as_matrix()
    t=rng.normal(size=3); t/=np.linalg.norm(t)
    X=rng.uniform(-1,1,size=(5,3))+np.array([0,0,4.0])
    b1=X/np.linalg.norm(X,axis=1,keepdims=True)
    X2=X@R.T+t; b2=X2/np.linalg.norm(X2,axis=1,keepdims=True)
    if len(sys.argv)>3 and trial % 2 == 1:   # contaminated sample: one or more correspondences replaced by random bearings
        nbadpts = 1 + trial % 4
        for q in range(nbadpts):
            v = rng.normal(size=3); v[2] = abs(v[2]) + 0.5; b2[q] = v / np.linalg.norm(v)
    b1=np.ascontiguousarray(b1); b2=np.ascontiguousarray(b2)
    Er=np.zeros(90); nr=C.c_int(0); ref.ref_five_point(P(b1),P(b2),P(Er),C.byref(nr))
    Ed=np.zeros(90); nd=C.c_int(0)
    rc=h.mvgx_debug_five_point(P(b1),P(b2),P(Ed),C.byref(nd))
    assert rc==0, h.mvgx_last_error()
    A=[norm(Er[9*k:9*k+9]) for k in range(nr.value)]; B=[norm(Ed[9*k:9*k+9]) for k in range(nd.value)]
    # match each reference model to the closest device model
    d=[min((np.abs(a-b).max() for b in B), default=9) for a in A]
    ok = nr.value==nd.value
    worst=max([worst]+d)
    if not ok or (d and max(d)>1e-8): nbad+=1; print(trial, 'n', nr.value, nd.value, 'maxdiff', max(d) if d else None)
    # the true E among them?
    Et=norm((np.array([[0,-t[2],t[1]],[t[2],0,-t[0]],[-t[1],t[0],0]])@R).ravel())
    dt=min(np.abs(Et-b).max() for b in B) if B else 9
    if dt>1e-8 and not (len(sys.argv)>3 and trial % 2 == 1): print(trial,'true E not found by device', dt, 'ref', min(np.abs(Et-a).max() for a in A) if A else 9)
fb=C.c_ulonglong(0); h.mvgx_debug_five_point_fallbacks(C.byref(fb), 0)
print('worst', worst, 'bad', nbad, 'hqr fallbacks', fb.value)
