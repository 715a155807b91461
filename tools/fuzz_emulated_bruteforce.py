import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmvg_amd import matching, synth
from tests import _emu, _oracle
from tests.test_l2u8_cpu import liop_like
rng = np.random.default_rng(3)
t0=time.time(); n=0; bad=0
pools = [list(range(0,6)), [62,63,64,65,66], [254,255,256,257,258], [511,512,513]]
while time.time()-t0 < 200:
    k = int(rng.integers(2,5))
    sizes = [int(rng.choice(pools[int(rng.integers(0,len(pools)))])) for _ in range(k)]
    pairs = np.array([(i,j) for i in range(k) for j in range(k) if i!=j], np.uint32)
    ratio = float(rng.choice([0.6,0.8,1.0]))
    kind = int(rng.integers(0,4)); bp = int(rng.choice([0,1,3])); seed=int(rng.integers(1<<30))
    with _emu.emulated():
        if kind == 0:
            L = int(rng.choice([64,32,61,20])); imgs = synth.binary_descriptors(k, sizes, n_bytes=L, seed=seed, flip_bits=max(2,L//2))
            o = _oracle.port_matcher_regions_match_hamming(imgs, pairs, ratio, L)
            ctx = matching.HammingContext(); 
            if bp: ctx.set_option("batch_pairs", bp)
            ctx.set_regions(imgs, L); _, off, ij = ctx.run(pairs, ratio); ctx.close()
        elif kind == 1:
            imgs = synth.float_descriptors(k, sizes, seed=seed)
            if rng.random() < 0.5 and min(sizes[:2]) > 1: imgs[1] = imgs[1].copy(); m=min(sizes[0],sizes[1]); imgs[1][:m:2] = imgs[0][:m:2]
            o = _oracle.port_matcher_regions_match_f32(imgs, pairs, ratio)
            ctx = matching.L2fContext()
            if bp: ctx.set_option("batch_pairs", bp)
            ctx.set_regions(imgs, 64); _, off, ij = ctx.run(pairs, np.float32(ratio)*np.float32(ratio)); ctx.close()
        else:
            dim = int(rng.choice([64,128,144])); imgs = liop_like(sizes, dim, seed=seed)
            o = _oracle.port_matcher_regions_match(imgs, pairs, ratio, dim=dim)
            ctx = matching.L2u8Context()
            if bp: ctx.set_option("batch_pairs", bp)
            ctx.set_regions(imgs, dim); _, off, ij = ctx.run(pairs, np.float32(ratio)*np.float32(ratio)); ctx.close()
    n += 1
    if not (np.array_equal(off,o[0]) and np.array_equal(ij,o[1])):
        bad += 1; print("MISMATCH", kind, sizes, ratio, bp, seed, flush=True)
print("cases", n, "mismatches", bad)
