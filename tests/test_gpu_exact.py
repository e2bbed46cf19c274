"""GPU: the exact-gradient / ADGRAPE functional path (SURVEY.md 8f-3 exact gradient, 8f-4 functional path):
grape_config.gradient = exact, objective = fom | c1, n <= 4; against the oracle's independent restatement
(block-triangular Pade), the mpmath fixtures, and the reference's four ADGRAPE testsets through solve()."""
import os

import numpy as np
import pytest

from conftest import assert_parity
from test_oracle_exact import EXACT, load_exact

pytestmark = pytest.mark.gpu
tol = 1e-6


@pytest.mark.parametrize("path", EXACT, ids=[os.path.basename(p)[:-5] for p in EXACT])
@pytest.mark.parametrize("objective", ["fom", "c1"])
@pytest.mark.parametrize("kernel", ["auto", "lane"])
def test_hip_exact_matches_mpmath(qoc, path, objective, kernel, monkeypatch):
    if kernel == "lane":
        monkeypatch.setenv("GRAPE_SMALL_KERNEL", "lane")
    c, A, B, Xi, Xt, wts, x = load_exact(path)
    exp = c["exact"]["objective0" if objective == "fom" else "objective1"]
    with qoc.GrapeEngine(c["sys_type"], A, B, Xi, Xt, wts, c["T"], c["N"], variant=c["variant"], gradient="exact",
                         objective=objective) as eng:
        F, G = eng.eval(x)
        foms, grads = eng.member_results()
    assert_parity(F, G, exp["F"], np.array(exp["G"]), c["n"], what="ensemble")
    for k in range(c["E"]):
        assert_parity(foms[k], grads[k], exp["member_F"][k], np.array(exp["member_g"][k]), c["n"], what=f"member {k}")


@pytest.mark.parametrize("name,kw", [("C3", {"E": 70, "N": 130}), ("C2", {"N": 333}), ("C1", {})])
@pytest.mark.parametrize("objective", ["fom", "c1"])
@pytest.mark.parametrize("variant", [0, 1])
def test_hip_exact_matches_oracle(qoc, oracle, name, kw, objective, variant):
    w = qoc.workloads.config(name, **kw)
    F_ref, G_ref = oracle.ensemble_exact(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T, variant=variant,
                                         objective=0 if objective == "fom" else 1)
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, variant=variant, gradient="exact",
                         objective=objective) as eng:
        F, G = eng.eval(w.x)
    assert_parity(F, G, F_ref, G_ref, w.n, what=f"{name} {objective} v{variant}")


@pytest.mark.parametrize("n,herm,sys_type,E,N", [(8, True, "StateTransfer", 3, 9), (16, False, "CoherenceTransfer", 2, 7),
                                                  (16, True, "UnitaryGate", 2, 6), (32, True, "UnitaryGate", 2, 4)])
@pytest.mark.parametrize("objective", ["fom", "c1"])
def test_hip_exact_tile_family_matches_oracle(qoc, oracle, n, herm, sys_type, E, N, objective):
    """the MFMA tile kernels' exact gradient (exact_tile.hip), n = 8, 16, 32."""
    rng = np.random.default_rng(7 * n + E)
    K = 2

    def gen():
        M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        return (M + M.conj().T) / 2 if herm else M
    A = np.array([gen() for _ in range(E)]) * 0.4
    B = np.array([[gen() for _ in range(K)] for _ in range(E)]) * 0.3

    def st():
        if sys_type == "UnitaryGate":
            return rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        v = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        v /= np.linalg.norm(v)
        return np.outer(v, v.conj())
    Xi, Xt = np.array([st() for _ in range(E)]), np.array([st() for _ in range(E)])
    if sys_type == "UnitaryGate":
        Xi, Xt = Xi / n, Xt / n
    wts = rng.uniform(0.3, 1.0, E)
    x = rng.uniform(-1, 1, (K, N))
    F_ref, G_ref = oracle.ensemble_exact(sys_type, A, B, Xi, Xt, wts, x, 1.2, variant=1, objective=0 if objective == "fom" else 1)
    with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.2, N, variant=1, gradient="exact", objective=objective) as eng:
        F, G = eng.eval(x)
    assert_parity(F, G, F_ref, G_ref, n, what=f"tile exact n={n} {objective}")


def test_hip_exact_vectorised_liouvillian(qoc):
    """16 x 1 vec(rho) under a dissipative Liouvillian (test/liou.jl's shape at two qubits), exact gradient of the
    C1 functional: the device with native n x 1 states against the 50-digit fixture (computed zero-padded)."""
    path = [p for p in EXACT if "vec_16x1" in p][0]
    c, A, B, Xi, Xt, wts, x = load_exact(path)
    exp = c["exact"]["objective1"]
    with qoc.GrapeEngine(c["sys_type"], A, B, Xi[:, :, :1], Xt[:, :, :1], wts, c["T"], c["N"], variant=c["variant"],
                         gradient="exact", objective="c1") as eng:
        F, G = eng.eval(x)
        assert eng.m == 1
    assert_parity(F, G, exp["F"], np.array(exp["G"]), c["n"], what="vec(rho) 16x1 exact")


def test_exact_with_squarings(qoc, oracle):
    """large dt |H|: the derivative through the scaling-and-squaring chain."""
    w = qoc.workloads.config("C3", E=3, N=8)
    w.T = 24.0
    F_ref, G_ref = oracle.ensemble_exact(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T, variant=1, objective=1)
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, variant=1, gradient="exact", objective="c1") as eng:
        F, G = eng.eval(w.x)
    assert_parity(F, G, F_ref, G_ref, w.n, what="exact, squaring path")


@pytest.mark.parametrize("name,kw", [("C3", {"E": 37, "N": 83}), ("C3", {"E": 3, "N": 500}), ("ref2x2", {}), ("ref2x2_pairs", {})])
@pytest.mark.parametrize("objective", ["fom", "c1"])
def test_exact_gradient_from_the_unitary_flow(qoc, oracle, monkeypatch, name, kw, objective):
    """UnitaryGate, Hermitian generators, lane-pair kernel: the exact gradient runs behind the UNITARY flow, whose backward
    sweep leaves W_t = X_t L_{t+1}' = M_t P_t' and tr M -- no debug flow, no X_t / L_t dumps (C3: 0.44 -> 0.28 ms).  Against
    the oracle, against the debug-flow path (GRAPE_EXACT_W1=0) and with the flags that keep the debug flow."""
    if name == "ref2x2_pairs":
        monkeypatch.setenv("GRAPE_SMALL_KERNEL", "pair")            # 2 x 2 on lane pairs: takes the new flow as well
    w = qoc.workloads.config(name, **kw) if name[:6] != "ref2x2" else qoc.workloads.reference_ensemble("UnitaryGate", 5, 25, 5.0)
    assert w.sys_type == "UnitaryGate"
    F_ref, G_ref = oracle.ensemble_exact(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T, variant=0,
                                         objective=0 if objective == "fom" else 1)
    args = (w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N)
    with qoc.GrapeEngine(*args, gradient="exact", objective=objective) as eng:
        F, G = eng.eval(w.x)
        flow_new = eng.info["unitary_flow"]
        if flow_new:
            with pytest.raises(qoc.GrapeError):
                eng.trajectory(0)                                   # this flow stores neither states nor costates
        P = eng.trajectory(0, states=False)[0]                      # ... but the propagators
    monkeypatch.setenv("GRAPE_EXACT_W1", "0")
    with qoc.GrapeEngine(*args, gradient="exact", objective=objective) as eng:
        F0, G0 = eng.eval(w.x)
        flow_old = eng.info["unitary_flow"]
        P0, X0 = eng.trajectory(0)
    monkeypatch.delenv("GRAPE_EXACT_W1")
    with qoc.GrapeEngine(*args, gradient="exact", objective=objective, flags=qoc.engine.FLAG_KEEP_COSTATES) as eng:
        Fk, Gk = eng.eval(w.x)
        assert eng.info["unitary_flow"] == 0
        eng.trajectory(0, costates=True)
    assert flow_old == 0 and flow_new == (1 if w.n == 4 or name == "ref2x2_pairs" else 0)   # (2 x 2 runs the lane kernel unless asked otherwise)
    assert_parity(F, G, F_ref, G_ref, w.n, what=f"{name} exact ({objective}) from the unitary flow")
    assert_parity(F0, G0, F_ref, G_ref, w.n, what=f"{name} exact ({objective}) from the debug flow")
    assert np.max(np.abs(G - G0)) <= 1e-10 * max(1.0, np.max(np.abs(G0))) and abs(F - F0) <= 1e-12
    assert np.array_equal(Gk, G0) and Fk == F0
    assert np.max(np.abs(P - P0)) <= 1e-13


def test_exact_argument_rules(qoc):
    w = qoc.workloads.config("C1")
    with pytest.raises(qoc.GrapeError) as ei:
        qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, objective="c1")     # C1 functional needs exact
    assert ei.value.status == -1


@pytest.mark.parametrize("name,kw,objective,devices", [("C3", {"E": 9, "N": 41}, "fom", None), ("C2", {"N": 57}, "c1", None),
                                                       ("C4", {"E": 3, "N": 10}, "c1", None), ("C3", {"E": 6, "N": 24}, "c1", [0, 0])])
def test_exact_gradient_in_batches(qoc, oracle, name, kw, objective, devices):
    """gradient = exact with max_batch > 1 (multi-start / the ladder's probes on the ADGRAPE path): entry b of the batch is
    bitwise what eval(X[b]) returns -- the arrays run one behind the other through the one stored trajectory -- on one
    device and on a group of shards."""
    w = qoc.workloads.config(name, **kw)
    rng = np.random.default_rng(3)
    X = np.stack([w.x, w.x + 0.1 * rng.standard_normal(w.x.shape), 0.5 * w.x])
    extra = dict(devices=devices, flags=qoc.engine.FLAG_GROUP_PEER_SUM) if devices else {}
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, variant=1, gradient="exact", objective=objective,
                         max_batch=3, **extra) as eng:
        Fb, Gb = eng.eval_batch(X)
        singles = [eng.eval(X[b]) for b in range(3)]
        F2, G2 = eng.eval_batch(X[:2])
    for b in range(3):
        F_ref, G_ref = oracle.ensemble_exact(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, X[b], w.T, variant=1,
                                             objective=0 if objective == "fom" else 1)
        assert_parity(Fb[b], Gb[b], F_ref, G_ref, w.n, what=f"{name} exact batch entry {b}")
        assert Fb[b] == singles[b][0] and np.array_equal(Gb[b], singles[b][1])
    assert np.array_equal(F2, Fb[:2]) and np.array_equal(G2, Gb[:2])


def _problem(qoc, sys_type, N, T):
    wl = qoc.workloads
    ug = sys_type == "UnitaryGate"
    return qoc.Problem(B=[wl.Sx, wl.Sy], A=wl.Sz, Xi=wl.U_init if ug else wl.rho_init,
                       Xt=wl.U_fin if ug else wl.rho_fin, T=T, n_controls=2, guess=wl.controls(2, N),
                       sys_type=qoc.UnitaryGate() if ug else qoc.StateTransfer())


@pytest.mark.parametrize("sys_type,N,floor", [("StateTransfer", 10, 0.75), ("UnitaryGate", 25, 0.0)])
@pytest.mark.parametrize("optimizer", ["host", "device"])
def test_reference_adgrape_single_testsets(qoc, sys_type, N, floor, optimizer):
    """test/state_transfer_tests.jl:103-118 and test/unitary_gate_tests.jl:115-132 (ADGRAPE, single problem)."""
    prob = _problem(qoc, sys_type, N, 1.0)
    sol = qoc.solve(prob, qoc.ADGRAPE(n_slices=N, optimizer=optimizer))
    assert isinstance(sol, qoc.SolutionResult)
    assert sol.result.minimum - floor < tol


@pytest.mark.parametrize("sys_type,N", [("StateTransfer", 25), ("UnitaryGate", 100)])
def test_reference_adgrape_ensemble_testsets(qoc, sys_type, N):
    """test/state_transfer_tests.jl:124-149 and test/unitary_gate_tests.jl:137-165 (ADGRAPE, n_ens = 5):
    `@test sol.result.minimum - C1(rho_fin, rho_fin) < tol`."""
    wl = qoc.workloads
    ug = sys_type == "UnitaryGate"
    prob = _problem(qoc, sys_type, N, 5.0)
    tgt = (wl.U_fin, wl.U_init) if ug else (wl.rho_fin, wl.rho_init)
    ens = qoc.EnsembleProblem(prob=prob, n_ens=5, A_g=lambda k: (k - 2.5) / 2.5 * wl.Sz * 5,
                              B_g=lambda k: [wl.Sx, wl.Sy], XiG=lambda k: prob.Xi,
                              XtG=lambda k: tgt[0] if k % 2 else tgt[1], wts=np.ones(5) / 5)
    sol = qoc.solve(ens, qoc.ADGRAPE(n_slices=N))
    assert isinstance(sol, qoc.EnsembleSolutionResult)
    assert sol.result.minimum - 0.75 < tol * 10
