"""smoke(): one tiny BA solve on cuda:0, checked against the oracle."""
from openmvg_amd import ba, synth
from tests import _oracle


def run():
    sc = synth.ba_scene(8, 120, track_len=5, model=3, seed=9)
    rc, osum, *_ = _oracle.port_ba_solve(sc)
    ctx = ba.BaContext(sc, device=0)
    s = ctx.solve()
    ctx.close()
    assert rc == 0 and abs(s.final_rmse - osum.final_rmse) < 1e-6, (s.final_rmse, osum.final_rmse)
    print(f"smoke BA ok: {s.num_iterations} LM iterations, RMSE {s.initial_rmse:.4f} -> {s.final_rmse:.6f} (oracle {osum.final_rmse:.6f})")
