"""The caller's choice of linear solver (VERDICT r3 missing #4): Bundle_Adjustment_Ceres::BA_Ceres_options::linear_solver_type_
(sfm_data_BA_ceres.cpp:132-146 default, :483 handed to ceres; sequential_SfM.cpp:1193-1205 picks DENSE_SCHUR / SPARSE_SCHUR by the pose
count). Library: mvgx_ba_set_linear_solver; replacement TU: DENSE_SCHUR -> dense Cholesky of the reduced camera system,
SPARSE_SCHUR -> the library's rule (block-sparse where its plan pays). Both must give the reference's answer."""
import ctypes as C

import numpy as np
import pytest

from openmvg_amd import _capi, ba, synth
from tests import _emu, _oracle


def _ring():   # (a scene the library's own rule solves block-sparse: test_ba_emu_cpu.test_block_sparse_solver_is_the_default...)
    return synth.ba_scene(n_cams=120, n_points=700, track_len=4, model=1, n_intr_groups=1, seed=72)


def _solve(sc, kind, iters=2, before=True):
    ctx = ba.BaContext(sc)
    try:
        if kind is not None and before:
            ctx.set_linear_solver(kind)
        s = ctx.solve(ba.default_options(max_num_iterations=iters))
        info = ctx.solver_info()
        return ctx, s, info, ctx.read_params()
    except Exception:
        ctx.close()
        raise


def _check_kinds(iters=2):
    sc = _ring()
    out = {}
    for kind in ("auto", "dense", "sparse", "sparse_preferred"):
        ctx, s, info, prm = _solve(sc, kind, iters=iters)
        out[kind] = (s, info.sparse, prm)
        # afterwards: naming the solver in place is fine, the other one is refused with MVGX_ERR_STATE and nothing changes
        ctx.set_linear_solver("dense" if not info.sparse else "sparse")
        with pytest.raises(_capi.MvgxError) as e:
            ctx.set_linear_solver("sparse" if not info.sparse else "dense")
        assert e.value.code == _capi.MVGX_ERR_STATE
        assert ctx.solver_info().sparse == info.sparse
        with pytest.raises(_capi.MvgxError) as e:
            ctx.set_linear_solver(7)
        assert e.value.code == _capi.MVGX_ERR_ARG
        ctx.close()
    assert [out[k][1] for k in ("auto", "dense", "sparse", "sparse_preferred")] == [1, 0, 1, 1]
    ref = out["dense"]
    for k in ("auto", "sparse", "sparse_preferred"):
        assert abs(out[k][0].final_cost - ref[0].final_cost) <= 1e-9 * ref[0].final_cost
        for a, b in zip(out[k][2], ref[2]):
            assert np.allclose(a, b, rtol=1e-9, atol=1e-9)
    return ref[0]


def test_set_linear_solver_emulated():
    with _emu.emulated():
        s = _check_kinds(iters=1)
    rc, osum, *_ = _oracle.port_ba_solve(_ring(), options=_oracle.default_ba_options(max_num_iterations=1))
    assert abs(s.final_cost - osum.final_cost) <= 1e-9 * osum.final_cost


def test_environment_outranks_the_call(monkeypatch):
    monkeypatch.setenv("MVGX_BA_SOLVER", "dense")
    with _emu.emulated():
        ctx, s, info, _ = _solve(_ring(), "sparse", iters=1)
        ctx.close()
    assert info.sparse == 0


def test_tiny_reduced_system_sparse_preferred():
    """one tile: the sparse plan is trivial; a request for it must still solve to the oracle's numbers"""
    sc = synth.ba_scene(n_cams=6, n_points=80, track_len=4, model=3, n_intr_groups=1, seed=5)
    rc, osum, opp, opi, opx, _ = _oracle.port_ba_solve(sc, options=_oracle.default_ba_options(max_num_iterations=3))
    with _emu.emulated():
        for kind in ("sparse_preferred", "dense"):
            ctx, s, info, (poses, intr, pts) = _solve(sc, kind, iters=3)
            ctx.close()
            assert abs(s.final_cost - osum.final_cost) <= 1e-9 * osum.final_cost and np.allclose(pts, opx, atol=1e-9)


def _kept_info(lib):
    info = _capi.BaSolverInfo()
    rc = lib.mvgx_adapter_ba_kept_solver_info(C.byref(info))
    return rc, info


def _adapter_route(lib, stats_fn, iters=3):
    """Adjust() through the replacement TU with DENSE_SCHUR, then SPARSE_SCHUR, then DENSE_SCHUR again on the same scene: the kept
    context is re-used only while the solver kind stays"""
    sc = _ring()
    lib.mvgx_adapter_ba_release_context()
    stats_fn(reset=True)
    got = {}
    for step, (ls, sparse) in enumerate([(1, 0), (1, 0), (2, 1), (1, 0)]):
        rc, stats, poses, intr, pts = _oracle.ref_ba_adjust_ex(sc, max_iterations=iters, linear_solver=ls, lib=lib)
        assert rc == 0 and stats[3] == 1.0
        krc, info = _kept_info(lib)
        assert krc == 0 and info.sparse == sparse, (step, ls, info.sparse)
        got[step] = (stats[1], pts)
    created, reused, subset = stats_fn()
    assert (created, reused) == (3, 1), (created, reused, subset)
    for step in (1, 2, 3):
        assert abs(got[step][0] - got[0][0]) < 1e-9 and np.allclose(got[step][1], got[0][1], atol=1e-8)
    lib.mvgx_adapter_ba_release_context()
    return sc, got[0]


def _stats3(lib):
    def f(reset=False):
        out = (C.c_uint64 * 3)()
        lib.mvgx_adapter_ba_context_stats3(out, 1 if reset else 0)
        return tuple(int(v) for v in out)
    return f


def test_replacement_tu_maps_linear_solver_type_emulated():
    lib = _oracle.adapter_ba_emu()
    if lib is None:
        pytest.skip("emulated adapter not built")
    sc, (rmse, pts) = _adapter_route(lib, _stats3(lib), iters=1)
    if _oracle.have_ref_ba():   # the reference itself, DENSE_SCHUR
        rc, stats, *_ = _oracle.ref_ba_adjust_ex(sc, max_iterations=1, linear_solver=1)
        assert abs(stats[1] - rmse) < 1e-6


@pytest.mark.gpu
def test_set_linear_solver_on_the_mi355x():
    _check_kinds(iters=3)


@pytest.mark.gpu
def test_replacement_tu_maps_linear_solver_type_on_the_mi355x():
    lib = _oracle.adapter()
    sc, (rmse, pts) = _adapter_route(lib, _stats3(lib))
    if _oracle.have_ref_ba():
        rc, stats, *_ = _oracle.ref_ba_adjust_ex(sc, max_iterations=3, linear_solver=1)
        assert abs(stats[1] - rmse) < 1e-6
