"""LM iteration time against the track length of the scene at a fixed number of observations (1M, 200 cameras): points with more than 10
observations leave the fused point-group path (kGroupCams) for the record-based one. Usage: ba_track_length.py [lengths...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmvg_amd import ba, synth
for tl in [int(a) for a in sys.argv[1:]] or [6, 10, 11, 16]:
    sc = synth.ba_scene(n_cams=200, n_points=1000000 // tl, track_len=tl, model=1, n_intr_groups=1, seed=0xBA5E0003)
    c = ba.BaContext(sc); c.solve(ba.default_options(max_num_iterations=3)); c.close()
    c = ba.BaContext(sc)
    s = c.solve(ba.default_options(max_num_iterations=4))
    print("track length", tl, "points", 1000000 // tl, "iterations", s.num_iterations, "iter_ms", round(s.iter_ms_mean, 3), "rmse", round(s.final_rmse, 4), flush=True)
    c.close()
