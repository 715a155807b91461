#!/usr/bin/env python
"""Multi-GPU readiness check: when more than one device is visible, runs the sharded forms of both halves of
the hot path inside THIS process and compares them with the one-device run:
  matching   64 images x 2000 descriptors, exhaustive pairs: mvgx_match_create_multi over all devices (dynamic batch sharing,
             no collective) against one device - the lists must be identical;
  BA         configs[2]-sized scene: mvgx_ba_create_multi over all devices through BOTH transports (RCCL all-reduce of the reduced
             camera system from one host thread per device - every rank's communicator passes the known-answer self-check of
             mvgx_comm.hip inside mvgx_ba_comm_init - and the peer-mapped sums over xGMI) against one device: same iteration
             counts, final RMSE equal to 1e-12.
Prints one JSON line; exit status 0 iff everything agreed (or only one device is visible: {"devices": 1, "skipped": true}).
bench.py --gpus N runs it from rank 0 before the timed region and copies the line into its record ("scale_selfcheck")."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from openmvg_amd import _capi, ba, matching, synth
    rehearsal = os.environ.get("MVGX_SELFCHECK_REHEARSAL") == "1"   # bench.py --rehearsal: emulated devices, tiny sizes, ordinals "0,0"
    if rehearsal:
        from tests import _emu
        globals()['_emulation'] = _emu.emulated()   # (kept alive: the context manager restores the real library when collected)
        globals()['_emulation'].__enter__()
        os.environ.setdefault("MVGX_SELFCHECK_DEVICES", "0,0")
    n_dev = _capi.device_count()
    out = {"devices": int(n_dev)}
    env = os.environ.get("MVGX_SELFCHECK_DEVICES")   # e.g. "0,0": the sharded forms with a repeated ordinal on a one-GPU box (peer transport only)
    devs = [int(x) for x in env.split(",")] if env else list(range(n_dev))
    if len(devs) < 2:
        out["skipped"] = True
        print(json.dumps(out))
        return 0
    out["ordinals"] = devs
    ok = True
    # ---- matching ----
    n_img, n_desc = (6, 80) if rehearsal else (64, 2000)
    descs = synth.image_descriptors(n_img, n_desc=n_desc, seed=0xC0FFEE00)
    pairs = matching.exhaustive_pairs_array(n_img)
    rsq = np.float32(0.8) * np.float32(0.8)
    c1 = matching.MatchContext(0); c1.set_regions(descs)
    _, off1, ij1 = c1.run(pairs, rsq); c1.close()
    t0 = time.perf_counter()
    cm = matching.MatchContext(devices=devs); cm.set_option("batch_pairs", 4 if rehearsal else 64); cm.set_regions(descs)
    _, offm, ijm = cm.run(pairs, rsq); cm.close()
    same = bool(np.array_equal(off1, offm) and np.array_equal(ij1, ijm))
    out["matching"] = {"image_pairs": int(len(pairs)), "matches": int(off1[-1]), "identical_to_one_device": same, "seconds": time.perf_counter() - t0}
    ok &= same
    # ---- BA, both transports ----
    import bench_ba
    scene = synth.ba_scene(**bench_ba.ba_config(1, rehearsal=rehearsal))
    if rehearsal:
        os.environ["MVGX_BA_MULTI_MIN_OBS"] = "1"
    c = ba.BaContext(scene, device=0); s1 = c.solve(); c.close()
    out["ba"] = {"one_device": {"iterations": int(s1.num_iterations), "final_rmse": float(s1.final_rmse)}}
    for transport in (("rccl", "peer") if len(set(devs)) == len(devs) else ("peer",)):   # RCCL needs distinct devices
        os.environ["MVGX_BA_TRANSPORT"] = transport
        rec = {}
        try:
            t0 = time.perf_counter()
            c = ba.BaContext(scene, devices=devs); s = c.solve(); c.close()
            rec = {"iterations": int(s.num_iterations), "final_rmse": float(s.final_rmse), "rmse_diff_vs_one_device": abs(float(s.final_rmse) - float(s1.final_rmse)),
                   "lm_iteration_ms": float(s.iter_ms_mean), "seconds": time.perf_counter() - t0,
                   "agrees": bool(s.num_iterations == s1.num_iterations and abs(s.final_rmse - s1.final_rmse) <= 1e-12)}
            if transport == "rccl":
                rec["rccl_ranks_self_checked"] = len(devs)   # mvgx_ba_comm_init fails the create if a rank's known-answer all-reduce is wrong
        except Exception as e:
            rec = {"agrees": False, "error": repr(e)}
        finally:
            os.environ.pop("MVGX_BA_TRANSPORT", None)
        out["ba"][transport] = rec
        ok &= bool(rec.get("agrees"))
    out["ok"] = bool(ok)
    print(json.dumps(out))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
