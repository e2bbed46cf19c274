"""GPU: the native 4 x 1 chain (sweep_vec4.hip; VERDICT r5 Missing #1) -- vec(rho) of one qubit under a Liouvillian, the evolution
test/liou.jl:38-48 of the reference writes out by hand: n = 4, states 4 x 1, UnitaryGate-style left multiplication.  The kernel
keeps the vector chains sequential (a chain wave per workgroup, DPP row = member) and hides them under the propagators' Taylor
series.  Opt-in (GRAPE_VEC4=1): see DESIGN.md section 8 for what it costs next to the lane-pair kernel.  Parity per member
against the oracle and the 50-digit fixtures (1e-10 bar), against the zero-padded run of the lane-pair kernel (1e-12), the
stored trajectory, tile edges (N = 55 / 56 / 57 / 113), ragged workgroups (E not a multiple of four), batches, both variants,
squarings, device L-BFGS -- and the contexts that must stay with the lane-pair kernel."""
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


@pytest.mark.parametrize("E,N,K,herm,scale", [(5, 57, 2, False, 0.6), (4, 56, 3, False, 0.6), (1, 55, 1, True, 0.6),
                                              (9, 113, 2, False, 0.6), (6, 40, 4, False, 3.0), (3, 200, 8, False, 0.4)])
@pytest.mark.parametrize("variant", [0, 1])
def test_vec4_matches_oracle_and_the_padded_run(qoc, oracle, monkeypatch, E, N, K, herm, scale, variant):
    A, B, Xi, Xt, wts, x = _random(E, N, K, seed=1000 + 10 * E + N, herm=herm, scale=scale)
    T = 1.5
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval("UnitaryGate", A, B, Xi, Xt, wts, x, T, variant=variant, per_member=True)
    res = {}
    for tag, env in (("vec4", "1"), ("padded", "0")):
        monkeypatch.setenv("GRAPE_VEC4", env)
        with qoc.GrapeEngine("UnitaryGate", A, B, Xi, Xt, wts, T, N, variant=variant, member_results=True) as eng:
            F, G = eng.eval(x)
            foms, grads = eng.member_results()
            names = eng.kernel_names()
            F2, G2 = eng.eval(x)
            if tag == "vec4":
                k = E - 1
                P, X = eng.trajectory(k)[:2]
        assert ("vec4_sweep_kernel" in names) == (tag == "vec4"), names
        assert F == F2 and np.array_equal(G, G2)             # bitwise reproducible
        assert_parity(F, G, F_ref, G_ref, 4, what=tag)
        for m in range(E):
            assert_parity(foms[m], grads[m], foms_ref[m], grads_ref[m], 4, what=f"{tag} member {m}")
        res[tag] = (F, G)
    assert abs(res["vec4"][0] - res["padded"][0]) <= 1e-12 * max(1.0, abs(res["padded"][0]))
    assert np.abs(res["vec4"][1] - res["padded"][1]).max() <= 1e-12 * max(1.0, np.abs(res["padded"][1]).max())
    _, _, P_ref, X_ref, _ = oracle.member_eval("UnitaryGate", A[k], B[k], Xi[k], Xt[k], x, T, variant=variant, trajectory=True)
    assert P.shape == (N, 4, 4) and X.shape == (N + 1, 4, 1)
    assert np.abs(P - P_ref).max() <= 2e-13 * max(1.0, np.abs(P_ref).max())
    assert np.abs(X - X_ref).max() <= 2e-13 * max(1.0, np.abs(X_ref).max())


@pytest.mark.parametrize("name", ["vec_4x1_liou_v0", "vec_4x1_liou_v1"])
def test_vec4_against_the_mpmath_fixtures(qoc, monkeypatch, name):
    """test/liou.jl's single-qubit Liouville-space evolution, 50-digit fixtures (tests/golden, oracle/make_golden.py)."""
    from test_oracle_golden import load_case
    monkeypatch.setenv("GRAPE_VEC4", "1")
    c, A, B, Xi, Xt, wts, x, exp, _ = load_case(os.path.join(os.path.dirname(__file__), "golden", name + ".json"))
    with qoc.GrapeEngine(c["sys_type"], A, B, Xi, Xt, wts, c["T"], c["N"], variant=c["variant"], member_results=True) as eng:
        F, G = eng.eval(x)
        foms, grads = eng.member_results()
        assert "vec4_sweep_kernel" in eng.kernel_names()
    assert_parity(F, G, exp["F"], np.array(exp["G"]), c["n"], what="ensemble")
    for k in range(c["E"]):
        assert_parity(foms[k], grads[k], exp["member_F"][k], np.array(exp["member_g"][k]), c["n"], what=f"member {k}")


def test_vec4_at_size_batches_and_lbfgs(qoc, oracle, monkeypatch):
    """E = 600, N = 130 (three tiles, ragged last one; 150 workgroups): spot members against the oracle for the dissipative
    Liouvillian AND the Hermitian one; batches; L-BFGS runs on such a context.  Without the switch the lane-pair kernel keeps
    the problem (the kernel is opt-in: it does not beat the unitary flow on the reference's own Hermitian case)."""
    monkeypatch.setenv("GRAPE_VEC4", "1")
    for diss in (True, False):
        w = qoc.workloads.liouville_vec(nq=1, E=600, N=130, T=2.0, dissipative=diss)
        with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, member_results=True, max_batch=3) as eng:
            F, G = eng.eval(w.x)
            foms, grads = eng.member_results()
            assert "vec4_sweep_kernel" in eng.kernel_names() and eng.info["kernel_family"] == 2
            xs = np.stack([w.x, 0.5 * w.x, -0.3 * w.x])
            Fb, Gb = eng.eval_batch(xs)
            assert Fb[0] == F and np.array_equal(Gb[0], G)
            F1, G1 = eng.eval(0.5 * w.x)
            assert Fb[1] == F1 and np.array_equal(Gb[1], G1)
            if diss:
                x_min, info = eng.lbfgs(w.x, iterations=6)
                assert info["minimum"] <= F + 1e-12
        for k in (0, 299, 599):
            f_ref, g_ref = oracle.member_eval(w.sys_type, w.A[k], w.B[k], w.Xi[k], w.Xt[k], w.x, w.T)
            assert_parity(foms[k], grads[k], f_ref, g_ref, 4, what=f"dissipative={diss} member {k}")
    monkeypatch.delenv("GRAPE_VEC4", raising=False)
    w = qoc.workloads.liouville_vec(nq=1, E=600, N=130, T=2.0, dissipative=True)
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N) as eng:
        eng.eval(w.x)
        assert "vec4_sweep_kernel" not in eng.kernel_names() and eng.info["kernel_family"] == 0


def test_vec4_leaves_other_contexts_alone(qoc, monkeypatch):
    """Stored costates, the exact gradient, 4 x 2 states and sandwich problems stay with the lane-pair kernel even when forced."""
    monkeypatch.setenv("GRAPE_VEC4", "1")
    A, B, Xi, Xt, wts, x = _random(3, 30, 2, seed=5)
    with qoc.GrapeEngine("UnitaryGate", A, B, Xi, Xt, wts, 1.0, 30, flags=qoc.engine.FLAG_KEEP_COSTATES) as eng:
        eng.eval(x)
        assert "vec4_sweep_kernel" not in eng.kernel_names()
        eng.trajectory(0, costates=True)
    with qoc.GrapeEngine("UnitaryGate", A, B, Xi, Xt, wts, 1.0, 30, gradient="exact") as eng:
        eng.eval(x)
        assert "vec4_sweep_kernel" not in eng.kernel_names()
    rng = np.random.default_rng(3)
    Xi2 = rng.standard_normal((3, 4, 2)) + 0j
    with qoc.GrapeEngine("UnitaryGate", A, B, Xi2, Xi2.copy(), wts, 1.0, 30) as eng:
        eng.eval(x)
        assert "vec4_sweep_kernel" not in eng.kernel_names()
