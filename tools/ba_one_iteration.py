"""Two LM iterations on the bench scene C5 (for kernel-timing experiments under rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_ba
from openmvg_amd import ba, synth
sc = synth.ba_scene(**bench_ba.ba_config(1, "c5"))
c = ba.BaContext(sc)
try:
    c.solve(ba.default_options(max_num_iterations=2))
except Exception as e:
    print("solve:", e)
c.close()
