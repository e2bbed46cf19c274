"""GPU: 4 x 1 states at n = 4 under left multiplication (VERDICT r5 Missing #1) -- vec(rho) of one qubit under a Liouvillian,
the evolution test/liou.jl:38-48 of the reference writes out by hand.  With a dissipator the generator is not Hermitian and the
lane-pair kernel takes the general flow; since round 6 its sweep back runs on VECTORS there (sweep_pair.hip, template
parameter VEC: forward pass x_j+1 = P_j x_j inside the chunk, w_j = P_j' w_j+1 backward, g = gs Re(z w' B' x); phase A stores no
in-chunk prefixes) -- `sweep_pair_vec_kernel`, chosen by itself; GRAPE_PAIR_VEC=0 keeps the sweep on the padded matrices.
Parity per member against the oracle and the 50-digit fixtures (1e-10 bar), against the padded sweep (1e-12), ragged chunks,
workgroups with fewer members than slots, batches, both variants, squarings, device L-BFGS, the fallback beyond 16 slices per lane
-- and the contexts that keep the matrix sweep (Hermitian generators: unitary flow; stored costates; the exact gradient; 4 x 2
states)."""
import os

import numpy as np
import pytest

from conftest import assert_parity

pytestmark = pytest.mark.gpu


def _random(n_members, N, K, seed, herm=False, scale=0.6):
    rng = np.random.default_rng(seed)

    def gen():
        M = rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4))
        return (M + M.conj().T) / 2 if herm else M
    A = np.array([gen() for _ in range(n_members)]) * scale
    B = np.array([[gen() for _ in range(K)] for _ in range(n_members)]) * scale * 0.6
    Xi = rng.standard_normal((n_members, 4, 1)) + 1j * rng.standard_normal((n_members, 4, 1))
    Xt = rng.standard_normal((n_members, 4, 1)) + 1j * rng.standard_normal((n_members, 4, 1))
    wts = rng.uniform(0.2, 1.0, n_members)
    x = rng.uniform(-1, 1, (K, N))
    return A, B, Xi, Xt, wts, x


@pytest.mark.parametrize("E,N,K,herm,scale", [(5, 57, 2, False, 0.6), (4, 64, 3, False, 0.6), (1, 55, 1, False, 0.6),
                                              (9, 113, 2, False, 0.6), (6, 40, 4, False, 3.0), (3, 1000, 8, False, 0.4),
                                              (2, 700, 2, False, 0.5), (3, 90, 2, True, 0.6)])
@pytest.mark.parametrize("variant", [0, 1])
def test_vector_sweep_matches_oracle_and_the_matrix_sweep(qoc, oracle, monkeypatch, E, N, K, herm, scale, variant):
    A, B, Xi, Xt, wts, x = _random(E, N, K, seed=1000 + 10 * E + N, herm=herm, scale=scale)
    T = 1.5
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval("UnitaryGate", A, B, Xi, Xt, wts, x, T, variant=variant, per_member=True)
    res = {}
    for tag, env in (("vectors", None), ("matrices", "0")):
        if env is None:
            monkeypatch.delenv("GRAPE_PAIR_VEC", raising=False)
        else:
            monkeypatch.setenv("GRAPE_PAIR_VEC", env)
        with qoc.GrapeEngine("UnitaryGate", A, B, Xi, Xt, wts, T, N, variant=variant, member_results=True) as eng:
            F, G = eng.eval(x)
            foms, grads = eng.member_results()
            names = eng.kernel_names()
            info = eng.info
            F2, G2 = eng.eval(x)
        # Hermitian generators take the unitary flow whatever the states are; the vector sweep serves the general flow
        assert ("sweep_pair_vec_kernel" in names) == (tag == "vectors" and not herm), (names, info["slices_per_lane"])
        assert info["unitary_flow"] == (1 if herm else 0) and info["slices_per_lane"] <= 16
        assert F == F2 and np.array_equal(G, G2)             # bitwise reproducible
        assert_parity(F, G, F_ref, G_ref, 4, what=tag)
        for m in range(E):
            assert_parity(foms[m], grads[m], foms_ref[m], grads_ref[m], 4, what=f"{tag} member {m}")
        res[tag] = (F, G)
    assert abs(res["vectors"][0] - res["matrices"][0]) <= 1e-12 * max(1.0, abs(res["matrices"][0]))
    assert np.abs(res["vectors"][1] - res["matrices"][1]).max() <= 1e-12 * max(1.0, np.abs(res["matrices"][1]).max())


@pytest.mark.parametrize("name", ["vec_4x1_liou_v0", "vec_4x1_liou_v1"])
def test_vector_sweep_against_the_mpmath_fixtures(qoc, monkeypatch, name):
    """test/liou.jl's single-qubit Liouville-space evolution, 50-digit fixtures (tests/golden, oracle/make_golden.py); FORCE_GENERAL
    sends the (Hermitian) fixture problems through the general flow, i.e. through the vector sweep."""
    from test_oracle_golden import load_case
    monkeypatch.delenv("GRAPE_PAIR_VEC", raising=False)
    c, A, B, Xi, Xt, wts, x, exp, _ = load_case(os.path.join(os.path.dirname(__file__), "golden", name + ".json"))
    with qoc.GrapeEngine(c["sys_type"], A, B, Xi, Xt, wts, c["T"], c["N"], variant=c["variant"], member_results=True,
                         flags=qoc.engine.FLAG_FORCE_GENERAL) as eng:
        F, G = eng.eval(x)
        foms, grads = eng.member_results()
        assert "sweep_pair_vec_kernel" in eng.kernel_names()
    assert_parity(F, G, exp["F"], np.array(exp["G"]), c["n"], what="ensemble")
    for k in range(c["E"]):
        assert_parity(foms[k], grads[k], exp["member_F"][k], np.array(exp["member_g"][k]), c["n"], what=f"member {k}")


def test_vector_sweep_at_size_batches_lbfgs_and_the_fallback(qoc, oracle, monkeypatch):
    """E = 600, N = 130 (150 workgroups, chunks of 3 slices with a ragged tail): spot members against the oracle, batches,
    L-BFGS; the Hermitian Liouvillian keeps the unitary flow; N = 2200 at E = 1024 gets eight waves per member (grape_create keeps
    lane pairs at 16 slices or fewer) and stays with the vector sweep; N = 4200 needs 17 slices per lane even then -- beyond the
    vector sweep's registers: the matrix sweep serves it."""
    monkeypatch.delenv("GRAPE_PAIR_VEC", raising=False)
    w = qoc.workloads.liouville_vec(nq=1, E=600, N=130, T=2.0, dissipative=True)
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, member_results=True, max_batch=3) as eng:
        F, G = eng.eval(w.x)
        foms, grads = eng.member_results()
        assert "sweep_pair_vec_kernel" in eng.kernel_names() and eng.info["kernel_family"] == 0
        xs = np.stack([w.x, 0.5 * w.x, -0.3 * w.x])
        Fb, Gb = eng.eval_batch(xs)
        assert Fb[0] == F and np.array_equal(Gb[0], G)
        F1, G1 = eng.eval(0.5 * w.x)
        assert Fb[1] == F1 and np.array_equal(Gb[1], G1)
        x_min, info = eng.lbfgs(w.x, iterations=6)
        assert info["minimum"] <= F + 1e-12
    for k in (0, 299, 599):
        f_ref, g_ref = oracle.member_eval(w.sys_type, w.A[k], w.B[k], w.Xi[k], w.Xt[k], w.x, w.T)
        assert_parity(foms[k], grads[k], f_ref, g_ref, 4, what=f"member {k}")
    w = qoc.workloads.liouville_vec(nq=1, E=600, N=130, T=2.0, dissipative=False)
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N) as eng:
        eng.eval(w.x)
        assert "sweep_pair_vec_kernel" not in eng.kernel_names() and eng.info["unitary_flow"] == 1
    for N, vec in ((2200, True), (4200, False)):
        w = qoc.workloads.liouville_vec(nq=1, E=1024, N=N, T=2.0, dissipative=True)
        with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, member_results=True) as eng:
            eng.eval(w.x)
            foms, grads = eng.member_results()
            assert (eng.info["slices_per_lane"] <= 16) == vec and ("sweep_pair_vec_kernel" in eng.kernel_names()) == vec
            assert eng.info["waves_per_member"] == 8
        f_ref, g_ref = oracle.member_eval(w.sys_type, w.A[7], w.B[7], w.Xi[7], w.Xt[7], w.x, w.T)
        assert_parity(foms[7], grads[7], f_ref, g_ref, 4, what=f"N = {N} member 7")


def test_vector_sweep_leaves_other_contexts_alone(qoc, monkeypatch):
    """Stored costates (debug flow), the exact gradient and 4 x 2 states keep the matrix sweep."""
    monkeypatch.delenv("GRAPE_PAIR_VEC", raising=False)
    A, B, Xi, Xt, wts, x = _random(3, 30, 2, seed=5)
    with qoc.GrapeEngine("UnitaryGate", A, B, Xi, Xt, wts, 1.0, 30, flags=qoc.engine.FLAG_KEEP_COSTATES) as eng:
        eng.eval(x)
        assert "sweep_pair_vec_kernel" not in eng.kernel_names()
        eng.trajectory(0, costates=True)
    with qoc.GrapeEngine("UnitaryGate", A, B, Xi, Xt, wts, 1.0, 30, gradient="exact") as eng:
        eng.eval(x)
        assert "sweep_pair_vec_kernel" not in eng.kernel_names()
    rng = np.random.default_rng(3)
    Xi2 = rng.standard_normal((3, 4, 2)) + 0j
    with qoc.GrapeEngine("UnitaryGate", A, B, Xi2, Xi2.copy(), wts, 1.0, 30) as eng:
        eng.eval(x)
        assert "sweep_pair_vec_kernel" not in eng.kernel_names()
