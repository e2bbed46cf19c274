"""HIP path vs the CPU oracle on identical seeded inputs, through the C ABI (ctypes).
Tolerance: the north star's 1e-10 relative (conftest.assert_parity)."""
import numpy as np
import pytest

from conftest import assert_parity

pytestmark = pytest.mark.gpu


def _engine(qoc, w, **kw):
    return qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, **kw)


CASES = [
    ("C1", {}, {}),
    ("C2", {}, {}),
    ("C2", {}, {"waves_per_member": 4}),
    ("C3", {"E": 8, "N": 40}, {}),
    ("C3", {"E": 8, "N": 500}, {}),
    ("C3", {"E": 8, "N": 500}, {"waves_per_member": 1}),
    ("C3", {"E": 5, "N": 130}, {"waves_per_member": 2, "slices_per_lane": 3}),
    ("C3", {"E": 64, "N": 500}, {}),
]


@pytest.mark.parametrize("name,wkw,ekw", CASES)
@pytest.mark.parametrize("variant", [0, 1])
def test_ensemble_parity(qoc, oracle, name, wkw, ekw, variant):
    w = qoc.workloads.config(name, **wkw)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(
        w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T, variant=variant, per_member=True)
    with _engine(qoc, w, variant=variant, **ekw) as eng:
        F, G = eng.eval(w.x)
        foms, grads = eng.member_results()
    for k in range(w.E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], w.n, what=f"{name} member {k}")
    assert_parity(F, G, F_ref, G_ref, w.n, what=name)


@pytest.mark.parametrize("sys_type", ["UnitaryGate", "StateTransfer"])
def test_trajectory_parity(qoc, oracle, sys_type):
    w = qoc.workloads.reference_ensemble(sys_type, n_ens=5, N=25, T=5.0)
    with _engine(qoc, w, flags=qoc.engine.FLAG_KEEP_COSTATES) as eng:
        eng.eval(w.x)
        for k in range(w.E):
            P, X, L = eng.trajectory(k, costates=True)
            _, _, Pr, Xr, Lr = oracle.member_eval(w.sys_type, w.A[k], w.B[k], w.Xi[k], w.Xt[k], w.x, w.T,
                                                  trajectory=True)
            assert np.abs(P - Pr).max() < 1e-13
            assert np.abs(X - Xr).max() < 1e-12
            assert np.abs(L - Lr).max() < 1e-12
