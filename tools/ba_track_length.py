"""LM iteration time against the track length of the scene at a fixed number of observations (1M, 200 cameras): points with more than 10
observations leave the fused point-group path (kGroupCams) for the record-based one. Usage: ba_track_length.py [lengths... | mixed]
"mixed": track lengths 2 + geometric, mean 6, tail to 40 (synth.geometric_track_lengths) at the same ~1M observations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmvg_amd import ba, synth
import numpy as np
for tl in [a if a == "mixed" else int(a) for a in sys.argv[1:]] or [6, 10, 11, 16, "mixed"]:
    if tl == "mixed":
        lens = synth.geometric_track_lengths(1000000 // 6, mean=6.0, lo=2, hi=40)
        sc = synth.ba_scene(n_cams=200, n_points=len(lens), track_lens=lens, model=1, n_intr_groups=1, seed=0xBA5E0003)
        print("mixed: observations", sc["n_obs"], "in tracks longer than 10:", int(lens[lens > 10].sum()), "points longer than 10:", int((lens > 10).sum()), "of", len(lens), flush=True)
    else:
        sc = synth.ba_scene(n_cams=200, n_points=1000000 // tl, track_len=tl, model=1, n_intr_groups=1, seed=0xBA5E0003)
    c = ba.BaContext(sc); c.solve(ba.default_options(max_num_iterations=3)); c.close()
    c = ba.BaContext(sc)
    s = c.solve(ba.default_options(max_num_iterations=4))
    print("track length", tl, "points", sc["n_points"], "observations", sc["n_obs"], "iterations", s.num_iterations, "iter_ms", round(s.iter_ms_mean, 3), "rmse", round(s.final_rmse, 4), flush=True)
    c.close()
