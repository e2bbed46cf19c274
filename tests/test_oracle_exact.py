"""The oracle's exact gradient / ADGRAPE functional restatement (oracle_member_exact) pinned three ways:
50-digit mpmath fixtures (tests/golden/exact, oracle/make_golden.py), central finite differences of the
objective itself, and consistency with the GRAPE figure of merit."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import assert_parity

EXACT = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "exact", "*.json")))


def load_exact(path):
    c = json.load(open(path))
    n = c["n"]

    def mat(lst):
        a = np.array(lst)
        return (a[:, 0] + 1j * a[:, 1]).reshape(n, n).T

    inp = c["inputs"]
    return (c, np.array([mat(m) for m in inp["A"]]), np.array([[mat(m) for m in bk] for bk in inp["B"]]),
            np.array([mat(m) for m in inp["Xi"]]), np.array([mat(m) for m in inp["Xt"]]), np.array(inp["wts"]),
            np.array(inp["x"]))


def test_exact_fixtures_present():
    assert len(EXACT) >= 3


@pytest.mark.parametrize("path", EXACT, ids=[os.path.basename(p)[:-5] for p in EXACT])
@pytest.mark.parametrize("objective", [0, 1])
def test_oracle_exact_matches_mpmath(oracle, path, objective):
    c, A, B, Xi, Xt, wts, x = load_exact(path)
    exp = c["exact"][f"objective{objective}"]
    F, G, foms, grads = oracle.ensemble_exact(c["sys_type"], A, B, Xi, Xt, wts, x, c["T"], variant=c["variant"],
                                              objective=objective, per_member=True)
    assert_parity(F, G, exp["F"], np.array(exp["G"]), c["n"], what="ensemble")
    for k in range(c["E"]):
        assert_parity(foms[k], grads[k], exp["member_F"][k], np.array(exp["member_g"][k]), c["n"], what=f"member {k}")


@pytest.mark.parametrize("name,kw", [("C3", {"E": 2, "N": 9}), ("C2", {"N": 12})])
@pytest.mark.parametrize("objective", [0, 1])
def test_exact_gradient_is_the_derivative_of_the_objective(oracle, qoc, name, kw, objective):
    """central differences of the objective, h = 1e-6: O(h^2) truncation + 1e-16/h rounding ~ 1e-9."""
    w = qoc.workloads.config(name, **kw)
    args = (w.sys_type, w.A[0], w.B[0], w.Xi[0], w.Xt[0])
    F, G = oracle.member_exact(*args, w.x, w.T, variant=1, objective=objective)
    h = 1e-6
    rng = np.random.default_rng(3)
    for _ in range(6):
        c, t = rng.integers(w.K), rng.integers(w.N)
        xp, xm = w.x.copy(), w.x.copy()
        xp[c, t] += h
        xm[c, t] -= h
        fd = (oracle.member_exact(*args, xp, w.T, 1, objective)[0] - oracle.member_exact(*args, xm, w.T, 1, objective)[0]) / (2 * h)
        assert abs(fd - G[c, t]) <= 2e-9 * max(1.0, np.abs(G).max()), (c, t, fd, G[c, t])


@pytest.mark.parametrize("name,kw", [("C3", {"E": 2, "N": 9}), ("C1", {})])
def test_objective_values(oracle, qoc, name, kw):
    """objective 0 IS the GRAPE figure of merit; objective 1 is C1(Xt, U Xi [U']) (src/solve.jl:272-276, :284-288)."""
    w = qoc.workloads.config(name, **kw)
    args = (w.sys_type, w.A[0], w.B[0], w.Xi[0], w.Xt[0])
    F_fom, _ = oracle.member_eval(*args, w.x, w.T, variant=1)
    F0, _ = oracle.member_exact(*args, w.x, w.T, 1, 0)
    F1, _ = oracle.member_exact(*args, w.x, w.T, 1, 1)
    assert abs(F0 - F_fom) <= 1e-13
    _, _, P, X, L = oracle.member_eval(*args, w.x, w.T, variant=1, trajectory=True)
    assert abs(F1 - oracle.C1(w.Xt[0], X[-1])) <= 1e-13
    if w.sys_type != "UnitaryGate":
        assert F1 == F0                                   # same functional for State/CoherenceTransfer
