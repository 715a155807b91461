#!/usr/bin/env python
"""Check the 64 x 64 factor-and-invert kernel (test hook mvgx_debug_factor64) against numpy.
Usage: factor64_check.py [path-to-lib]   (default: the product library; pass the emulation library to run on CPU)"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests import _factor64

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "openmvg_amd", "lib", "libmvgx_hip.so")
lib = ctypes.CDLL(os.path.abspath(path))
worst = 0.0
for seed, kb in enumerate((64, 64, 64, 37, 16, 5)):
    e = _factor64.factor64_errors(lib, kb, seed)
    print("kb", kb, "L err %.2e" % e[0], "Linv err %.2e / %.2e" % (e[1], e[2]))
    worst = max(worst, *e)
print("worst", worst)
sys.exit(0 if worst < 1e-10 else 1)
