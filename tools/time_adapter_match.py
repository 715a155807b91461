"""End-to-end wall time of Matcher_Regions::Match (container included): the reference TU vs the MI355X replacement TU, same
caller code (oracle/ref_shim_match.cpp::ref_matcher_regions_match_u8_timed)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmvg_amd import synth
from tests import _oracle

def run(lib, n_images):
    descs = synth.image_descriptors(n_images, n_desc=2000, seed=0xC0FFEE00)
    arrs, ptrs, cnt = _oracle._desc_tables(descs)
    out = np.zeros(3)
    lib.ref_matcher_regions_match_u8_timed.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_float, C.c_void_p]
    lib.ref_matcher_regions_match_u8_timed(ptrs, cnt, n_images, C.c_float(0.8), out.ctypes.data)
    return out

ad = C.CDLL(_oracle.ADAPTER_SO, mode=os.RTLD_LOCAL | os.RTLD_NOW)
for n in (60, 300, 1000):
    o = run(ad, n)
    print(f"replacement: {n} images x 2000: Match() {o[0]:.3f} s, {int(o[1])} matches in {int(o[2])} pairs, {n*(n-1)/2*4e6/o[0]:.3e} descriptor pairs/s", flush=True)
if _oracle.have_ref_match():
    o = run(_oracle.ref_match(), 60)
    print(f"reference:   60 images x 2000: Match() {o[0]:.3f} s, {int(o[1])} matches in {int(o[2])} pairs, {60*59/2*4e6/o[0]:.3e} descriptor pairs/s", flush=True)
