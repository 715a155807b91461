"""Shared helpers of the geometric-filter tests: the parity policy in code.

Policy: the device path and the restatement reproduce the reference's sample sequence, tables, NFA arithmetic and
control flow exactly, but take the null space of the seven-point system from another factorisation than Eigen's eigen-solver, so a
fundamental matrix agrees with the reference's to rounding only. A pair therefore either (a) ends with the SAME inlier set - then
its NFA must be equal to 1e-9 relative, its precision equal and its F equal to 1e-6 after normalisation - or (b) belongs to the small
share whose decisive residual lies within rounding of a histogram edge / whose cubic is badly conditioned in one basis; that
share is counted and bounded (the compiled reference shows the same sensitivity to its own Eigen build)."""
import numpy as np


def normalised(F):
    F = np.asarray(F, np.float64).reshape(-1, 3, 3)
    F = F / np.linalg.norm(F, axis=(1, 2), keepdims=True)
    f = F.reshape(len(F), -1)
    return F * np.sign(f[np.arange(len(F)), np.abs(f).argmax(1)])[:, None, None]


def compare(start, ref, got_mask, got_ok, got_F, got_prec, got_nfa):
    """Returns (pairs_differing, report). Asserts policy (a) on the pairs whose inlier sets agree."""
    start = np.asarray(start, np.int64)
    n_pairs = len(start) - 1
    differing = []
    for p in range(n_pairs):
        lo, hi = start[p], start[p + 1]
        if bool(ref["ok"][p]) != bool(got_ok[p]) or not np.array_equal(ref["mask"][lo:hi], got_mask[lo:hi]):
            differing.append(p)
    same = np.ones(n_pairs, bool)
    same[differing] = False
    both_ok = same & np.asarray(ref["ok"], bool)
    if both_ok.any():
        dF = np.abs(normalised(ref["F"][both_ok]) - normalised(np.asarray(got_F)[both_ok])).max()
        assert dF < 1e-6, f"F differs by {dF} on pairs with identical inlier sets"
        assert np.allclose(ref["precision"][both_ok], np.asarray(got_prec)[both_ok], rtol=1e-12, atol=0)
        assert np.allclose(ref["nfa"][both_ok], np.asarray(got_nfa)[both_ok], rtol=1e-9, atol=1e-9)
    return differing, {"pairs": n_pairs, "pairs_ok_reference": int(np.asarray(ref["ok"]).sum()), "pairs_differing": len(differing)}
