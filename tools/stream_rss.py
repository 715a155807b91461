"""Host memory of a streamed run: N images x 2000 descriptors, all pairs through
mvgx_match_run_stream; the resident set is sampled at every batch. The same pairs then go through mvgx_match_run in chunks of
131 072 pairs (the collecting entry point) and the per-pair match counts + an order-sensitive checksum of the lists must agree.
Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import psutil
from openmvg_amd import matching, synth

n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
descs = synth.image_descriptors(n_images, n_desc=2000, seed=0xC0FFEE00)
pairs = matching.exhaustive_pairs_array(n_images)
r2 = np.float32(0.8) * np.float32(0.8)
ctx = matching.MatchContext(0)
ctx.set_regions(descs)
proc = psutil.Process()
W = np.uint64(0x9E3779B97F4A7C15)


def checksum(first_pair, off, lists):
    # order-sensitive inside a batch, additive across batches: sum over matches of (global match index + 1) * (i * 2^32 + j) * W
    if not len(lists):
        return np.uint64(0)
    v = (lists[:, 0].astype(np.uint64) << np.uint64(32)) | lists[:, 1].astype(np.uint64)
    pair_of = np.repeat(np.arange(len(off) - 1, dtype=np.uint64) + np.uint64(first_pair), np.diff(off.astype(np.int64)))
    rank_in_pair = np.arange(len(v), dtype=np.uint64) - np.repeat(off[:-1].astype(np.uint64), np.diff(off.astype(np.int64)))
    with np.errstate(over="ignore"):
        return np.sum((pair_of * np.uint64(1000003) + rank_in_pair + np.uint64(1)) * (v * W + np.uint64(12345)), dtype=np.uint64)


ctx.run_stream(pairs[:65536], r2)    # buffers allocated, python heap warmed
rss = []
counts_s = np.zeros(len(pairs), np.uint32)
state = {"sum": np.uint64(0), "matches": 0, "max_batch_bytes": 0}


def on_batch(p0, off, lists):
    counts_s[p0:p0 + len(off) - 1] = np.diff(off.astype(np.int64))
    with np.errstate(over="ignore"):
        state["sum"] = state["sum"] + checksum(p0, off, lists)
    state["matches"] += len(lists)
    state["max_batch_bytes"] = max(state["max_batch_bytes"], lists.nbytes)
    rss.append(proc.memory_info().rss)


rss0 = proc.memory_info().rss
t0 = time.perf_counter()
st = ctx.run_stream(pairs, r2, on_batch)
dt = time.perf_counter() - t0
rss1 = proc.memory_info().rss
# the collecting entry point, chunk by chunk
counts_c = np.zeros(len(pairs), np.uint32)
sum_c = np.uint64(0)
CH = 131072
for p0 in range(0, len(pairs), CH):
    _, off, ij = ctx.run(pairs[p0:p0 + CH], r2)
    counts_c[p0:p0 + len(off) - 1] = np.diff(off.astype(np.int64))
    B = 32768   # same batch seams as the streamed run, so that the checksum terms line up
    for b0 in range(0, len(off) - 1, B):
        o = off[b0:b0 + B + 1]
        with np.errstate(over="ignore"):
            sum_c = sum_c + checksum(p0 + b0, (o - o[0]).astype(np.uint32), ij[int(o[0]):int(o[-1])])
ctx.close()
print(json.dumps({
    "images": n_images, "image_pairs": int(len(pairs)), "matches": int(state["matches"]), "list_bytes_of_the_run": int(state["matches"]) * 8,
    "largest_batch_bytes": int(state["max_batch_bytes"]), "rss_before": int(rss0), "rss_after": int(rss1), "rss_max_during": int(max(rss)),
    "rss_growth_bytes": int(max(rss) - rss0), "seconds_with_python_sink": dt, "descriptor_pairs_per_s_with_python_sink": float(st.n_desc_pairs) / dt,
    "identical_counts": bool(np.array_equal(counts_s, counts_c)), "identical_checksum": bool(state["sum"] == sum_c)}))
