"""HIP path vs the CPU oracle on identical seeded inputs, through the C ABI (ctypes).
Tolerance: the north star's 1e-10 relative (conftest.assert_parity)."""
import numpy as np
import pytest

from conftest import assert_parity

pytestmark = pytest.mark.gpu


def _engine(qoc, w, **kw):
    return qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, member_results=True, **kw)


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
@pytest.mark.parametrize("flow", ["unitary", "general"])
def test_ensemble_parity(qoc, oracle, name, wkw, ekw, variant, flow):
    """both data flows of the sweep kernel: the Hermitian-generator (unitary) flow that carries
    M_t = P' M P backwards, and the general flow that stores forward states like the reference."""
    w = qoc.workloads.config(name, **wkw)
    ekw = dict(ekw, flags=0 if flow == "unitary" else qoc.engine.FLAG_FORCE_GENERAL)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(
        w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T, variant=variant, per_member=True)
    with _engine(qoc, w, variant=variant, **ekw) as eng:
        F, G = eng.eval(w.x)
        foms, grads = eng.member_results()
        assert eng.info["unitary_flow"] == (1 if flow == "unitary" else 0)
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


@pytest.mark.parametrize("sys_type", ["UnitaryGate", "StateTransfer"])
def test_non_hermitian_generator_takes_general_flow(qoc, oracle, sys_type):
    """a damped (non-Hermitian) drift: propagators are not unitary, so the library must pick the
    general flow by itself and still match the oracle."""
    w = qoc.workloads.config("C3", E=6, N=90)
    w.sys_type = sys_type
    w.A = w.A - 0.3j * np.diag([0.0, 0.2, 0.5, 1.0])[None]
    if sys_type == "StateTransfer":
        rho = np.zeros((4, 4), complex); rho[0, 0] = 1
        w.Xi = np.broadcast_to(rho, w.Xi.shape).copy()
        w.Xt = np.broadcast_to(np.full((4, 4), 0.25 + 0j), w.Xt.shape).copy()
    F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T)
    with _engine(qoc, w) as eng:
        F, G = eng.eval(w.x)
        assert eng.info["unitary_flow"] == 0
    assert_parity(F, G, F_ref, G_ref, w.n, what="damped")
