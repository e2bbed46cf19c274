"""GPU: rank-one states with member-invariant control operators -- the evaluation on vectors alone
(csrc/action_thin.hip: exp(G_t) applied to the two chains' vectors by its Taylor series, no propagators) -- against
the oracle's DENSE evaluation of the same inputs (the reference's formulas, src/GRAPE.jl:216-303, with
exp(-i dt H) from the Pade expm) at the 1e-10 bar, and against the library's expm + vector-chain flow (GRAPE_ACTION=0)."""
import numpy as np
import pytest

from conftest import assert_parity

pytestmark = pytest.mark.gpu


def _problem(n, K, N, E, sand, herm_gen, herm_ctrl, seed, scale=1.0, shared=True):
    rng = np.random.default_rng(seed)

    def gen(h):
        M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        return (M + M.conj().T) / 2 if h else M
    A = np.array([gen(herm_gen) for _ in range(E)]) * 0.6 * scale
    B0 = np.array([gen(herm_ctrl) for _ in range(K)]) * 0.4 * scale
    B = np.array([B0 if shared else np.array([gen(herm_ctrl) for _ in range(K)]) * 0.4 * scale for _ in range(E)])

    def vec():
        v = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        return v / np.linalg.norm(v)
    if sand:
        def rho():
            v = vec()
            return np.outer(v, v.conj())
        Xi = np.array([rho() for _ in range(E)])
        Xt = np.array([rho() for _ in range(E)])
    else:
        Xi = np.array([vec().reshape(n, 1) for _ in range(E)])
        Xt = np.array([vec().reshape(n, 1) for _ in range(E)])
    return A, B, Xi, Xt, rng.uniform(0.2, 1.0, E), rng.uniform(-1, 1, (K, N))


CASES = [  # n, K, N, E, sys_type, Hermitian generators (drift), Hermitian controls
    (16, 4, 1, 2, "CoherenceTransfer", False, True), (16, 4, 2, 2, "CoherenceTransfer", False, True),
    (16, 3, 3, 3, "StateTransfer", True, True), (16, 2, 4, 2, "CoherenceTransfer", False, False),
    (16, 4, 5, 2, "StateTransfer", False, False), (12, 1, 7, 3, "CoherenceTransfer", False, True),
    (9, 6, 8, 2, "StateTransfer", False, False), (16, 5, 33, 2, "CoherenceTransfer", False, True),
    (13, 3, 100, 5, "StateTransfer", True, True), (16, 4, 257, 3, "CoherenceTransfer", False, True),
    (16, 4, 1, 2, "UnitaryGate", False, True), (16, 3, 2, 3, "UnitaryGate", True, True),
    (10, 2, 5, 2, "UnitaryGate", False, False), (16, 6, 64, 4, "UnitaryGate", False, False),
    (16, 4, 101, 2, "UnitaryGate", False, True), (15, 2, 130, 70, "CoherenceTransfer", False, True),
    # n = 5..8: contexts that pack two members per tile; the vector flow works on members
    (5, 2, 9, 3, "CoherenceTransfer", False, True), (8, 3, 40, 5, "StateTransfer", True, False),
    (6, 2, 17, 4, "UnitaryGate", False, True), (8, 4, 64, 2, "UnitaryGate", True, True), (7, 1, 3, 1, "CoherenceTransfer", False, True),
]


@pytest.mark.parametrize("n,K,N,E,sys_type,herm_gen,herm_ctrl", CASES)
@pytest.mark.parametrize("variant", [0, 1])
def test_vector_flow_matches_dense_oracle(qoc, oracle, monkeypatch, n, K, N, E, sys_type, herm_gen, herm_ctrl, variant):
    monkeypatch.setenv("GRAPE_ACTION", "1")
    sand = sys_type != "UnitaryGate"
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, sand, herm_gen, herm_ctrl, seed=7 * n + N + K)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, 1.2, variant=variant,
                                                             per_member=True)
    with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.2, N, variant=variant, member_results=True) as eng:
        info = eng.info
        assert info["rank_one_chain"] == 1 and info["expm_action"] == 1 and info["time_chunks"] == 0
        assert info["fused_forward"] == 0 and info["states_stored"] == 0
        F, G = eng.eval(x)
        foms, grads = eng.member_results()
        F2, G2 = eng.eval(x)
        assert F == F2 and np.array_equal(G, G2)          # run-to-run bitwise
        with pytest.raises(qoc.engine.GrapeError, match="forms no propagators"):
            eng.trajectory(0, states=False)
    for k in range(E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"member {k}")
    assert_parity(F, G, F_ref, G_ref, n, what="ensemble")
    # the library's expm kernel + vector chain on the same inputs
    monkeypatch.setenv("GRAPE_ACTION", "0")
    with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.2, N, variant=variant) as eng:
        assert eng.info["rank_one_chain"] == (1 if n >= 9 else 0) and eng.info["expm_action"] == 0    # n <= 8: dense chains
        F_c, G_c = eng.eval(x)
    assert_parity(F, G, F_c, G_c, n, what="vector flow vs expm + chain")


@pytest.mark.parametrize("n,K,N,E,sys_type,herm_gen,herm_ctrl", [c for c in CASES if c[3] != 4 or c[0] != 16])
@pytest.mark.parametrize("whole", ["0", "1"])
def test_both_wave_layouts_of_the_shared_controls_kernel(qoc, oracle, monkeypatch, n, K, N, E, sys_type, herm_gen, herm_ctrl, whole):
    """action_parts_kernel<false> (one member per wave, DPP row = (direction, component)) and <true> (two members per
    wave, whole rows per lane; odd ensembles repeat the last member in the spare rows), forced by GRAPE_ACT_WHOLE -- the
    library picks by ensemble size (two per wave from 4 x compute units + 1 members on), which the small cases above never
    reach.  Two members of a wave share the Taylor plan (the larger norm bound), so member results may differ in the
    last bits between the layouts; both are held to the oracle."""
    monkeypatch.setenv("GRAPE_ACTION", "1")
    monkeypatch.setenv("GRAPE_ACT_WHOLE", whole)
    sand = sys_type != "UnitaryGate"
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, sand, herm_gen, herm_ctrl, seed=11 * n + N + 3 * K)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, 0.9, per_member=True)
    with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 0.9, N, member_results=True) as eng:
        assert eng.info["expm_action"] == 1
        F, G = eng.eval(x)
        foms, grads = eng.member_results()
        assert eng.kernel_names()[:2] == ["action_rows_kernel", "action_parts_kernel"]
        F2, G2 = eng.eval(x)
        assert F == F2 and np.array_equal(G, G2)
    for k in range(E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"member {k} (whole={whole})")
    assert_parity(F, G, F_ref, G_ref, n, what=f"ensemble (whole={whole})")


@pytest.mark.parametrize("whole", ["0", "1"])
def test_wave_layouts_with_large_generators(qoc, oracle, monkeypatch, whole):
    """degrees beyond 8 and generators in pieces (the slow path of the kernel's slice loop) on both layouts, with an odd
    ensemble whose members differ in norm by 30 x (two members of a wave share the larger plan)."""
    monkeypatch.setenv("GRAPE_ACTION", "1")
    monkeypatch.setenv("GRAPE_ACT_WHOLE", whole)
    n, K, N, E = 16, 3, 12, 5
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, True, False, True, seed=99)
    A = A * np.array([0.1, 3.0, 0.2, 9.0, 1.0])[:, None, None]
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval("CoherenceTransfer", A, B, Xi, Xt, wts, x, 2.0, per_member=True)
    with qoc.GrapeEngine("CoherenceTransfer", A, B, Xi, Xt, wts, 2.0, N, member_results=True) as eng:
        assert eng.info["expm_action"] == 1
        F, G = eng.eval(x)
        foms, grads = eng.member_results()
    for k in range(E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"member {k} (whole={whole})")
    assert_parity(F, G, F_ref, G_ref, n, what=f"ensemble (whole={whole})")


@pytest.mark.parametrize("scale,N,n,sys_type", [(0.01, 40, 16, "CoherenceTransfer"), (0.2, 40, 16, "CoherenceTransfer"),
                                                (3.0, 24, 16, "CoherenceTransfer"), (3.0, 24, 11, "UnitaryGate"),
                                                (12.0, 16, 16, "StateTransfer"), (40.0, 8, 16, "UnitaryGate")])
def test_degrees_and_pieces(qoc, oracle, monkeypatch, scale, N, n, sys_type):
    """dt |H| from 0.005 to ~100: Taylor degrees 4..20 from the table, then the generator split into pieces."""
    monkeypatch.setenv("GRAPE_ACTION", "1")
    K, E = 3, 3
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, sys_type != "UnitaryGate", True, True, seed=int(10 * scale) + N,
                                    scale=scale)
    F_ref, G_ref = oracle.ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, 1.0)
    with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.0, N) as eng:
        assert eng.info["expm_action"] == 1
        F, G = eng.eval(x)
    assert_parity(F, G, F_ref, G_ref, n, what=f"scale {scale}")


@pytest.mark.parametrize("squarings", [0, 3])
def test_forced_squarings_become_pieces(qoc, oracle, monkeypatch, squarings):
    monkeypatch.setenv("GRAPE_ACTION", "1")
    n, K, N, E = 16, 2, 20, 2
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, True, False, True, seed=4, scale=0.5)
    F_ref, G_ref = oracle.ensemble_eval("CoherenceTransfer", A, B, Xi, Xt, wts, x, 1.0)
    with qoc.GrapeEngine("CoherenceTransfer", A, B, Xi, Xt, wts, 1.0, N, expm_squarings=squarings) as eng:
        assert eng.info["expm_action"] == 1
        F, G = eng.eval(x)
    assert_parity(F, G, F_ref, G_ref, n, what=f"s = {squarings}")


def test_chosen_for_ensembles_that_fill_the_device(qoc, oracle, monkeypatch):
    # default threshold: one wavefront per member fills the SIMDs (4 x compute units members); lowered here to keep the
    # problem small
    monkeypatch.delenv("GRAPE_ACTION", raising=False)
    monkeypatch.setenv("GRAPE_ACTION_MIN", "256")
    n, K, N = 16, 2, 24
    for E, want in ((4, 0), (255, 0), (300, 1)):
        A, B, Xi, Xt, wts, x = _problem(n, K, N, E, True, False, True, seed=E)
        with qoc.GrapeEngine("CoherenceTransfer", A, B, Xi, Xt, wts, 1.0, N, member_results=True) as eng:
            assert eng.info["rank_one_chain"] == 1 and eng.info["expm_action"] == want
            F, G = eng.eval(x)
            foms, grads = eng.member_results()
        for k in (0, E // 2, E - 1):
            f_ref, g_ref = oracle.member_eval("CoherenceTransfer", A[k], B[k], Xi[k], Xt[k], x, 1.0)[:2]
            assert_parity(foms[k], grads[k], f_ref, g_ref, n, what=f"E = {E}, member {k}")
        assert abs(F - float(np.dot(wts, foms))) <= 1e-12 * max(1.0, abs(F))


OWN_CASES = [  # n, K, N, E, sys_type, Hermitian generators, Hermitian controls -- every member its own control operators
    (16, 4, 3, 3, "CoherenceTransfer", False, True), (16, 1, 1, 2, "StateTransfer", True, True),
    (12, 6, 33, 2, "CoherenceTransfer", False, False), (9, 2, 100, 5, "StateTransfer", False, True),
    (16, 3, 64, 4, "UnitaryGate", False, False), (13, 5, 7, 3, "UnitaryGate", True, True),
    (7, 2, 20, 3, "CoherenceTransfer", False, True), (16, 2, 130, 66, "CoherenceTransfer", False, True),
]


@pytest.mark.parametrize("n,K,N,E,sys_type,herm_gen,herm_ctrl", OWN_CASES)
@pytest.mark.parametrize("variant", [0, 1])
def test_vector_flow_with_per_member_controls(qoc, oracle, monkeypatch, n, K, N, E, sys_type, herm_gen, herm_ctrl, variant):
    """Members with their own control operators (n <= 16, at most six): the lane keeps its rows of them and forms the
    control sum itself -- no pre-pass; seven and more controls stay on the expm flow."""
    monkeypatch.setenv("GRAPE_ACTION", "1")
    sand = sys_type != "UnitaryGate"
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, sand, herm_gen, herm_ctrl, seed=5 * n + N + K, shared=False)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, 1.1, variant=variant,
                                                             per_member=True)
    with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.1, N, variant=variant, member_results=True) as eng:
        assert eng.info["expm_action"] == 1 and eng.info["hoisted_controls"] == 0
        F, G = eng.eval(x)
        foms, grads = eng.member_results()
    for k in range(E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"member {k}")
    assert_parity(F, G, F_ref, G_ref, n, what="ensemble")


def test_many_per_member_controls_keep_the_expm_flow(qoc, oracle, monkeypatch):
    monkeypatch.setenv("GRAPE_ACTION", "1")
    n, K, N, E = 16, 7, 12, 2
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, True, False, True, seed=2, shared=False)
    F_ref, G_ref = oracle.ensemble_eval("CoherenceTransfer", A, B, Xi, Xt, wts, x, 1.0)
    with qoc.GrapeEngine("CoherenceTransfer", A, B, Xi, Xt, wts, 1.0, N) as eng:
        assert eng.info["rank_one_chain"] == 1 and eng.info["expm_action"] == 0
        F, G = eng.eval(x)
    assert_parity(F, G, F_ref, G_ref, n, what="K = 7, own controls")


def test_uploads_switch_between_shared_and_per_member_controls(qoc, oracle, monkeypatch):
    monkeypatch.setenv("GRAPE_ACTION", "1")
    n, K, N, E = 16, 3, 30, 3
    A, B, Xi, Xt, wts, _ = _problem(n, K, N, E, True, False, True, seed=1)
    with qoc.GrapeEngine("CoherenceTransfer", A, B, Xi, Xt, wts, 1.0, N) as eng:
        for shared in (True, False, True):
            A, B, Xi, Xt, wts, x = _problem(n, K, N, E, True, False, True, seed=11, shared=shared)
            eng.set_operators(A, B, Xi, Xt, wts)
            assert eng.info["rank_one_chain"] == 1 and eng.info["expm_action"] == 1
            F, G = eng.eval(x)
            F_ref, G_ref = oracle.ensemble_eval("CoherenceTransfer", A, B, Xi, Xt, wts, x, 1.0)
            assert_parity(F, G, F_ref, G_ref, n, what=f"shared={shared}")


def test_batched_and_device_entry_points(qoc, oracle, monkeypatch):
    import torch
    monkeypatch.setenv("GRAPE_ACTION", "1")
    n, K, N, E = 16, 4, 50, 3
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, True, False, True, seed=3)
    rng = np.random.default_rng(0)
    xs = np.stack([x, rng.uniform(-1, 1, (K, N)), 0.5 * x])
    with qoc.GrapeEngine("CoherenceTransfer", A, B, Xi, Xt, wts, 1.0, N, max_batch=3) as eng:
        assert eng.info["expm_action"] == 1
        Fs, Gs = eng.eval_batch(xs)
        xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda")
        fg = torch.zeros(K * N + 1, dtype=torch.float64, device="cuda")
        eng.eval_device(xd.data_ptr(), fg.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        h = fg.cpu().numpy()
    for b in range(3):
        F_ref, G_ref = oracle.ensemble_eval("CoherenceTransfer", A, B, Xi, Xt, wts, xs[b], 1.0)
        assert_parity(Fs[b], Gs[b], F_ref, G_ref, n, what=f"batch {b}")
    assert_parity(h[-1], h[:-1].reshape(N, K).T, Fs[0], Gs[0], n, what="device entry point")


def test_nan_controls_propagate(qoc, monkeypatch):
    monkeypatch.setenv("GRAPE_ACTION", "1")
    n, K, N, E = 16, 2, 12, 2
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, True, False, True, seed=9)
    x[1, 5] = np.nan
    with qoc.GrapeEngine("CoherenceTransfer", A, B, Xi, Xt, wts, 1.0, N) as eng:
        F, G = eng.eval(x)
    assert np.isnan(F) and np.isnan(G).any()


@pytest.mark.parametrize("n,K,N,E,sys_type,herm_ctrl,shared", [
    (16, 3, 70, 3, "CoherenceTransfer", True, True), (16, 2, 64, 2, "StateTransfer", False, True),
    (12, 4, 17, 4, "UnitaryGate", True, False), (9, 1, 129, 2, "CoherenceTransfer", False, False),
    (16, 5, 1, 3, "UnitaryGate", False, True), (13, 2, 200, 5, "StateTransfer", True, True)])
def test_dense_forms_on_the_matrix_cores(qoc, oracle, monkeypatch, n, K, N, E, sys_type, herm_ctrl, shared):
    """Dense control operators, n <= 16: the bilinear forms as 16 x 16 products on the matrix cores
    (action_forms_mfma_kernel: four 16-slice tiles per wavefront, ragged last tiles) against the oracle and against the
    vector-ALU kernel (GRAPE_FORMS_VALU=1), on the Taylor flow and on the propagator chain."""
    sand = sys_type != "UnitaryGate"
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, sand, False, herm_ctrl, seed=3 * n + N, shared=shared)
    F_ref, G_ref = oracle.ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, 1.1)
    for flow in ("taylor", "propagators"):
        monkeypatch.setenv("GRAPE_ACTION", "1" if flow == "taylor" else "0")
        monkeypatch.setenv("GRAPE_THIN_DPP", "1")
        monkeypatch.setenv("GRAPE_HOIST", "1")
        res = []
        for valu in (False, True):
            if valu:
                monkeypatch.setenv("GRAPE_FORMS_VALU", "1")
            else:
                monkeypatch.delenv("GRAPE_FORMS_VALU", raising=False)
            with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.1, N) as eng:
                assert eng.info["expm_action"] == (1 if flow == "taylor" else 0) and eng.info["rank_one_chain"] == 1
                res.append(eng.eval(x))
            assert_parity(res[-1][0], res[-1][1], F_ref, G_ref, n, what=f"{flow}, vector-ALU forms={valu}")
        assert_parity(res[0][0], res[0][1], res[1][0], res[1][1], n, what="matrix-core forms vs vector-ALU forms")
    monkeypatch.delenv("GRAPE_FORMS_VALU", raising=False)


def _pauli_string(rng, nq=4):
    P = [np.eye(2), np.array([[0, 1], [1, 0]]), np.array([[0, -1j], [1j, 0]]), np.array([[1, 0], [0, -1]])]
    M = np.array([[1.0 + 0j]])
    for _ in range(nq):
        M = np.kron(M, P[int(rng.integers(0, 4))])
    return M


@pytest.mark.parametrize("terms,herm_ctrl,sys_type", [(1, True, "CoherenceTransfer"), (2, True, "CoherenceTransfer"),
                                                      (2, False, "CoherenceTransfer"), (3, True, "StateTransfer"),
                                                      (4, False, "StateTransfer"), (2, True, "UnitaryGate"),
                                                      (5, False, "UnitaryGate"), (6, True, "CoherenceTransfer"),
                                                      (7, True, "CoherenceTransfer")])
def test_sparse_control_operators_take_the_list_kernel(qoc, oracle, monkeypatch, terms, herm_ctrl, sys_type):
    """Controls that are sums of `terms` Pauli strings (at most `terms` non-zeros per row): the forms kernel reads
    (value, column) lists (1, 2, 3, 4 or 6 per row); 7 and more, or GRAPE_FORMS_DENSE, take the dense kernel."""
    monkeypatch.setenv("GRAPE_ACTION", "1")
    n, K, N, E = 16, 3, 70, 3
    rng = np.random.default_rng(100 * terms + herm_ctrl)
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, sys_type != "UnitaryGate", False, True, seed=terms)
    B0 = []
    for _ in range(K):
        M = sum(rng.uniform(0.2, 1.0) * _pauli_string(rng) for _ in range(terms))
        if not herm_ctrl:
            M = M * (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))      # same pattern, no symmetry
        B0.append(0.4 * M)
    B = np.array([B0] * E)
    F_ref, G_ref = oracle.ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, 1.0)
    res = []
    for dense in (False, True):
        if dense:
            monkeypatch.setenv("GRAPE_FORMS_DENSE", "1")
        with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.0, N) as eng:
            assert eng.info["expm_action"] == 1
            res.append(eng.eval(x))
        assert_parity(res[-1][0], res[-1][1], F_ref, G_ref, n, what=f"{terms} terms, dense={dense}")
    assert_parity(res[0][0], res[0][1], res[1][0], res[1][1], n, what="list kernel vs dense kernel")


def test_liouville_space_ensemble(qoc, oracle, monkeypatch):
    """C4's shape at test size: two-qubit Liouvillians with detuned members, vec(rho) states, shared control superoperators."""
    from quoptimalcontrol_jl_amd import workloads
    monkeypatch.setenv("GRAPE_ACTION", "1")
    w = workloads.config("C4", E=6, N=120)
    F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T)
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N) as eng:
        assert eng.info["expm_action"] == 1 and eng.info["sparse_controls"] == 1
        F, G = eng.eval(w.x)
    assert_parity(F, G, F_ref, G_ref, w.n, what="C4 at E = 6, N = 120")


CASES32 = [  # n, K, N, E, sys_type, Hermitian generators (drift), Hermitian controls
    (32, 3, 1, 2, "CoherenceTransfer", False, True), (32, 2, 2, 3, "StateTransfer", True, True),
    (17, 2, 5, 2, "CoherenceTransfer", False, False), (24, 4, 33, 3, "StateTransfer", False, True),
    (32, 6, 64, 2, "CoherenceTransfer", False, True), (31, 1, 70, 5, "CoherenceTransfer", True, False),
    (32, 3, 2, 2, "UnitaryGate", False, True), (20, 2, 9, 3, "UnitaryGate", True, True),
    (32, 4, 101, 2, "UnitaryGate", False, False), (29, 3, 40, 9, "UnitaryGate", False, True),
]


@pytest.mark.parametrize("n,K,N,E,sys_type,herm_gen,herm_ctrl", CASES32)
@pytest.mark.parametrize("variant", [0, 1])
def test_vector_flow_17_to_32(qoc, oracle, monkeypatch, n, K, N, E, sys_type, herm_gen, herm_ctrl, variant):
    """n = 17..32 (two tiles per side): rank-one states run on vectors where the vector flow applies; there is no
    expm-based vector chain at these sizes, so GRAPE_ACTION=0 is the dense chain."""
    monkeypatch.setenv("GRAPE_ACTION", "1")
    sand = sys_type != "UnitaryGate"
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, sand, herm_gen, herm_ctrl, seed=3 * n + N + K)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, 0.7, variant=variant,
                                                             per_member=True)
    with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 0.7, N, variant=variant, member_results=True) as eng:
        info = eng.info
        assert info["rank_one_chain"] == 1 and info["expm_action"] == 1 and info["kernel_family"] == 1
        F, G = eng.eval(x)
        foms, grads = eng.member_results()
        F2, G2 = eng.eval(x)
        assert F == F2 and np.array_equal(G, G2)
    for k in range(E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"member {k}")
    assert_parity(F, G, F_ref, G_ref, n, what="ensemble")
    monkeypatch.setenv("GRAPE_ACTION", "0")
    with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 0.7, N, variant=variant) as eng:
        assert eng.info["rank_one_chain"] == 0 and eng.info["expm_action"] == 0
        F_c, G_c = eng.eval(x)
    assert_parity(F, G, F_c, G_c, n, what="vector flow vs dense chain")


@pytest.mark.parametrize("scale,N", [(0.02, 20), (2.0, 12), (30.0, 6)])
def test_degrees_and_pieces_32(qoc, oracle, monkeypatch, scale, N):
    monkeypatch.setenv("GRAPE_ACTION", "1")
    n, K, E = 32, 2, 2
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, True, True, True, seed=int(10 * scale) + N, scale=scale)
    F_ref, G_ref = oracle.ensemble_eval("CoherenceTransfer", A, B, Xi, Xt, wts, x, 1.0)
    with qoc.GrapeEngine("CoherenceTransfer", A, B, Xi, Xt, wts, 1.0, N) as eng:
        assert eng.info["expm_action"] == 1
        F, G = eng.eval(x)
    assert_parity(F, G, F_ref, G_ref, n, what=f"scale {scale}")


def test_sparse_controls_32_and_selection(qoc, oracle, monkeypatch):
    """Five-qubit Pauli controls (one non-zero per row), n x 1 states; chosen without the switch from the threshold on,
    the dense chains below it."""
    monkeypatch.delenv("GRAPE_ACTION", raising=False)
    monkeypatch.setenv("GRAPE_ACTION_MIN", "40")
    n, K, N = 32, 3, 30
    rng = np.random.default_rng(5)
    for E, want in ((6, 0), (48, 1)):
        A, B, Xi, Xt, wts, x = _problem(n, K, N, E, False, True, True, seed=E)
        B0 = np.array([0.5 * _pauli_string(rng, nq=5) for _ in range(K)])
        B = np.array([B0] * E)
        with qoc.GrapeEngine("UnitaryGate", A, B, Xi, Xt, wts, 1.0, N, member_results=True) as eng:
            assert eng.info["expm_action"] == want and eng.info["rank_one_chain"] == want
            F, G = eng.eval(x)
            foms, grads = eng.member_results()
        for k in (0, E - 1):
            f_ref, g_ref = oracle.member_eval_rect(A[k], B[k], Xi[k], Xt[k], x, 1.0)[:2]
            assert_parity(foms[k], grads[k], f_ref, g_ref, n, what=f"E = {E}, member {k}")


@pytest.mark.parametrize("name", ["vec_16x1_diss_v0", "vec_32x1_5q_v0"])
def test_vector_flow_against_the_mpmath_fixtures(qoc, monkeypatch, name):
    """The 50-digit fixtures with n x 1 states and shared controls (tests/golden, oracle/make_golden.py): 16 x 1 vec(rho)
    under a dissipative Liouvillian, 32 x 1 under the five-qubit operators of C5."""
    import os
    from test_oracle_golden import load_case
    monkeypatch.setenv("GRAPE_ACTION", "1")
    c, A, B, Xi, Xt, wts, x, exp, _ = load_case(os.path.join(os.path.dirname(__file__), "golden", name + ".json"))
    with qoc.GrapeEngine(c["sys_type"], A, B, Xi, Xt, wts, c["T"], c["N"], variant=c["variant"], member_results=True) as eng:
        assert eng.info["expm_action"] == 1
        F, G = eng.eval(x)
        foms, grads = eng.member_results()
    assert_parity(F, G, exp["F"], np.array(exp["G"]), c["n"], what="ensemble")
    for k in range(c["E"]):
        assert_parity(foms[k], grads[k], exp["member_F"][k], np.array(exp["member_g"][k]), c["n"], what=f"member {k}")


def test_device_lbfgs_on_the_vector_flow(qoc, monkeypatch):
    """grape_lbfgs (Hager-Zhang, batched ladder) drives the vector flow like any other context: C4's operators at test
    size; F decreases monotonically and the returned point evaluates to the reported minimum."""
    from quoptimalcontrol_jl_amd import workloads
    monkeypatch.setenv("GRAPE_ACTION", "1")
    w = workloads.config("C4", E=5, N=80)
    for ls in ("hagerzhang", "ladder"):
        with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, max_batch=4) as eng:
            assert eng.info["expm_action"] == 1
            F0, _ = eng.eval(0.2 * w.x)
            x_min, info = eng.lbfgs(0.2 * w.x, iterations=12, line_search=ls)
            F_min, _ = eng.eval(x_min)
        assert info["iterations"] >= 3 and F_min < F0 - 1e-6          # (a nearly flat landscape: 1 - overlap^2 of generic states)
        assert abs(F_min - info["minimum"]) <= 1e-12


def test_amplitude_scaled_controls(qoc, oracle, monkeypatch):
    """A robustness ensemble over control-amplitude errors: B_kc = s_k B_c (C4's sparse superoperators) -- per-member
    operators with per-member (value, column) lists in the forms kernel."""
    from quoptimalcontrol_jl_amd import workloads
    monkeypatch.setenv("GRAPE_ACTION", "1")
    w = workloads.config("C4", E=5, N=90)
    scale = np.random.default_rng(3).uniform(0.8, 1.2, w.E)
    B = w.B * scale[:, None, None, None]
    F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, B, w.Xi, w.Xt, w.wts, w.x, w.T)
    for dense in (False, True):
        if dense:
            monkeypatch.setenv("GRAPE_FORMS_DENSE", "1")
        with qoc.GrapeEngine(w.sys_type, w.A, B, w.Xi, w.Xt, w.wts, w.T, w.N) as eng:
            assert eng.info["expm_action"] == 1 and eng.info["hoisted_controls"] == 0
            F, G = eng.eval(w.x)
        assert_parity(F, G, F_ref, G_ref, w.n, what=f"scaled controls, dense forms = {dense}")


PROP_CASES = [  # n, K, N, E, sys_type, Hermitian generators, Hermitian controls, shared controls
    (16, 4, 1, 2, "CoherenceTransfer", False, True, True), (16, 3, 2, 3, "StateTransfer", True, True, False),
    (16, 2, 6, 2, "CoherenceTransfer", False, False, True), (12, 8, 7, 3, "CoherenceTransfer", False, True, False),
    (9, 1, 8, 2, "StateTransfer", False, False, True), (16, 5, 33, 9, "CoherenceTransfer", False, True, True),
    (13, 3, 100, 5, "UnitaryGate", True, True, False), (16, 7, 64, 4, "UnitaryGate", False, False, False),
    (16, 4, 257, 3, "UnitaryGate", False, True, True), (15, 2, 130, 70, "CoherenceTransfer", False, True, True),
]


@pytest.mark.parametrize("n,K,N,E,sys_type,herm_gen,herm_ctrl,shared", PROP_CASES)
@pytest.mark.parametrize("variant", [0, 1])
def test_propagator_chain_on_dpp_products(qoc, oracle, monkeypatch, n, K, N, E, sys_type, herm_gen, herm_ctrl, shared, variant):
    """Ensembles below the Taylor flow's threshold (or with many per-member controls): the expm kernel's propagators
    (P_t and P_t^T dumps) and chain_prop_kernel -- one DPP matrix-vector product per slice, operands through an LDS-DMA
    ring -- then the same forms kernels.  Forced here for the small shapes."""
    monkeypatch.setenv("GRAPE_ACTION", "0")
    monkeypatch.setenv("GRAPE_THIN_DPP", "1")
    monkeypatch.setenv("GRAPE_HOIST", "1")
    sand = sys_type != "UnitaryGate"
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, sand, herm_gen, herm_ctrl, seed=11 * n + N + K, shared=shared)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, 1.2, variant=variant,
                                                             per_member=True)
    with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.2, N, variant=variant, member_results=True, max_batch=2) as eng:
        info = eng.info
        assert info["rank_one_chain"] == 1 and info["expm_action"] == 0 and info["time_chunks"] == 0 and info["fused_forward"] == 0
        F, G = eng.eval(x)
        foms, grads = eng.member_results()
        F2, G2 = eng.eval(x)
        assert F == F2 and np.array_equal(G, G2)
        P = eng.trajectory(E - 1, states=False)[0]          # this flow does form the propagators
        Fb, Gb = eng.eval_batch(np.stack([x, 0.5 * x]))
    for k in range(E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"member {k}")
    assert_parity(F, G, F_ref, G_ref, n, what="ensemble")
    assert_parity(Fb[0], Gb[0], F_ref, G_ref, n, what="batch entry 0")
    if sand:
        P_ref = oracle.member_eval(sys_type, A[-1], B[-1], Xi[-1], Xt[-1], x, 1.2, variant=variant, trajectory=True)[2]
    else:
        P_ref = oracle.member_eval_rect(A[-1], B[-1], Xi[-1], Xt[-1], x, 1.2, variant=variant, trajectory=True)[2]
    assert np.abs(P - P_ref).max() <= 1e-12 * max(1.0, np.abs(P_ref).max())
    monkeypatch.setenv("GRAPE_THIN_DPP", "0")
    with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.2, N, variant=variant) as eng:
        F_c, G_c = eng.eval(x)
    assert_parity(F, G, F_c, G_c, n, what="DPP chain vs sweep_thin.hip")


CHUNK_CASES = [  # n, K, N, E, sys_type, Hermitian generators, Hermitian controls, shared controls, chunks (None: the library's choice)
    (16, 4, 64, 1, "CoherenceTransfer", False, True, True, None), (16, 4, 1000, 1, "CoherenceTransfer", False, True, True, None),
    (16, 3, 100, 2, "StateTransfer", True, True, False, 3), (12, 2, 65, 3, "CoherenceTransfer", False, False, True, 32),
    (9, 1, 257, 5, "UnitaryGate", False, False, True, 2), (16, 7, 130, 4, "UnitaryGate", False, False, False, 13),
    (13, 3, 200, 40, "StateTransfer", False, True, True, None), (15, 2, 130, 33, "CoherenceTransfer", False, True, True, None),
    (16, 2, 71, 1, "UnitaryGate", True, True, True, 10), (16, 5, 99, 9, "CoherenceTransfer", False, True, True, 7),
]


@pytest.mark.parametrize("n,K,N,E,sys_type,herm_gen,herm_ctrl,shared,chunks", CHUNK_CASES)
@pytest.mark.parametrize("variant", [0, 1])
def test_propagator_chain_on_a_chunked_time_axis(qoc, oracle, monkeypatch, n, K, N, E, sys_type, herm_gen, herm_ctrl, shared,
                                                 chunks, variant):
    """Fewer than 80 rank-one members, down to ONE problem (src/solve.jl:63-143): expm kernel, chunk products, then
    chain_prop_kernel on the chunk products (the vectors at the chunk boundaries) and on the propagators with a workgroup
    per (member, chunk) -- ragged last chunks, chunks shorter than the operand ring, two chunks."""
    for name in ("GRAPE_ACTION", "GRAPE_THIN_DPP", "GRAPE_HOIST"):
        monkeypatch.delenv(name, raising=False)
    monkeypatch.setenv("GRAPE_DPP_CHUNKS", "1")              # (dense control operators here: not the library's choice)
    if chunks:
        monkeypatch.setenv("GRAPE_TP_CHUNKS", str(chunks))
    sand = sys_type != "UnitaryGate"
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, sand, herm_gen, herm_ctrl, seed=7 * n + N + K, shared=shared)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, 1.2, variant=variant,
                                                             per_member=True)
    with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.2, N, variant=variant, member_results=True, max_batch=2) as eng:
        info = eng.info
        assert info["rank_one_chain"] == 1 and info["prop_chain"] == 1 and info["expm_action"] == 0 and info["time_chunks"] >= 2
        F, G = eng.eval(x)
        foms, grads = eng.member_results()
        F2, G2 = eng.eval(x)
        assert F == F2 and np.array_equal(G, G2)
        P = eng.trajectory(E - 1, states=False)[0]
        Fb, Gb = eng.eval_batch(np.stack([0.5 * x, x]))
    for k in range(E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"member {k}")
    assert_parity(F, G, F_ref, G_ref, n, what="ensemble")
    assert_parity(Fb[1], Gb[1], F_ref, G_ref, n, what="batch entry 1")
    if sand:
        P_ref = oracle.member_eval(sys_type, A[-1], B[-1], Xi[-1], Xt[-1], x, 1.2, variant=variant, trajectory=True)[2]
    else:
        P_ref = oracle.member_eval_rect(A[-1], B[-1], Xi[-1], Xt[-1], x, 1.2, variant=variant, trajectory=True)[2]
    assert np.abs(P - P_ref).max() <= 1e-12 * max(1.0, np.abs(P_ref).max())
    monkeypatch.setenv("GRAPE_DPP_CHUNKS", "0")
    with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.2, N, variant=variant) as eng:
        assert eng.info["prop_chain"] == 0
        F_c, G_c = eng.eval(x)
    assert_parity(F, G, F_c, G_c, n, what="chunked propagator chain vs the round-2 flows")


@pytest.mark.parametrize("terms,herm_ctrl,sys_type,E,N,chunks,shared", [
    (1, True, "CoherenceTransfer", 1, 1000, None, True), (2, True, "CoherenceTransfer", 3, 200, None, True),
    (2, False, "CoherenceTransfer", 2, 130, 3, True), (3, True, "StateTransfer", 5, 64, 2, False),
    (4, False, "StateTransfer", 1, 257, 9, True), (2, True, "UnitaryGate", 4, 99, 33, False),
    (6, False, "UnitaryGate", 2, 71, None, True), (7, True, "CoherenceTransfer", 2, 100, None, True)])
def test_chunked_chain_with_sparse_controls(qoc, oracle, monkeypatch, terms, herm_ctrl, sys_type, E, N, chunks, shared):
    """A handful of members with sparse control operators (sums of `terms` Pauli strings; shared or the members' own) on the
    chunked propagator chain: the (value, column) list forms kernels and the dense one read the records of every chunk."""
    for name in ("GRAPE_ACTION", "GRAPE_THIN_DPP", "GRAPE_HOIST", "GRAPE_DPP_CHUNKS"):
        monkeypatch.delenv(name, raising=False)
    if chunks:
        monkeypatch.setenv("GRAPE_TP_CHUNKS", str(chunks))
    n, K = 16, 3
    rng = np.random.default_rng(10 * terms + herm_ctrl + E)
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, sys_type != "UnitaryGate", False, True, seed=terms + N)

    def controls():
        out = []
        for _ in range(K):
            M = sum(rng.uniform(0.2, 1.0) * _pauli_string(rng) for _ in range(terms))
            if not herm_ctrl:
                M = M * (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
            out.append(0.4 * M)
        return out
    B0 = controls()
    B = np.array([B0 if shared else controls() for _ in range(E)])
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, 1.0, per_member=True)
    res = []
    for dense in (False, True):
        if dense:
            monkeypatch.setenv("GRAPE_FORMS_DENSE", "1")
        if dense or terms > 6:                               # (the library picks this flow for at most six non-zeros per row)
            monkeypatch.setenv("GRAPE_DPP_CHUNKS", "1")
        with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.0, N, member_results=True, max_batch=2) as eng:
            assert eng.info["prop_chain"] == 1 and eng.info["time_chunks"] >= 2
            res.append(eng.eval(x))
            foms, grads = eng.member_results()
            for k in range(E):
                assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"member {k}, dense={dense}")
            Fb, Gb = eng.eval_batch(np.stack([x, -x]))
            assert_parity(Fb[0], Gb[0], F_ref, G_ref, n, what="batch entry 0")
        assert_parity(res[-1][0], res[-1][1], F_ref, G_ref, n, what=f"{terms} terms, dense={dense}")
    assert_parity(res[0][0], res[0][1], res[1][0], res[1][1], n, what="list forms kernel vs dense forms kernel")


@pytest.mark.parametrize("flow", ["taylor", "propagators", "chunked"])
@pytest.mark.parametrize("controls", ["sparse", "dense"])
def test_single_problem_is_closed_by_the_forms_kernel(qoc, oracle, monkeypatch, flow, controls):
    """ONE rank-one problem on a flow that ends in a forms kernel: that kernel writes the weighted [G, F] and its last
    workgroup publishes it (no reduce launch) -- bitwise what the reduce kernel hands out (GRAPE_DIRECT_PUBLISH=0), on the
    host path and through the device entry point."""
    import torch
    n, K, N = 16, 3, 150
    A, B, Xi, Xt, wts, x = _problem(n, K, N, 1, True, False, True, seed=21)
    if controls == "sparse":
        rng = np.random.default_rng(8)
        B = np.array([[0.3 * (_pauli_string(rng) + _pauli_string(rng)) for _ in range(K)]])
    wts = np.array([0.37])
    monkeypatch.setenv("GRAPE_ACTION", "1" if flow == "taylor" else "0")
    monkeypatch.setenv("GRAPE_HOIST", "1")
    if flow != "chunked":
        monkeypatch.setenv("GRAPE_THIN_DPP", "1")
        monkeypatch.setenv("GRAPE_DPP_CHUNKS", "0")
        monkeypatch.setenv("GRAPE_THIN_SINGLE", "1")              # (one problem: keep the vector flows)
    F_ref, G_ref = oracle.ensemble_eval("CoherenceTransfer", A, B, Xi, Xt, wts, x, 1.0)
    res = []
    for direct in ("1", "0"):
        monkeypatch.setenv("GRAPE_DIRECT_PUBLISH", direct)
        with qoc.GrapeEngine("CoherenceTransfer", A, B, Xi, Xt, wts, 1.0, N, member_results=True) as eng:
            info = eng.info
            assert info["expm_action"] == (1 if flow == "taylor" else 0)
            if flow == "chunked":
                assert info["prop_chain"] == 1 and info["time_chunks"] >= 2
            F, G = eng.eval(x)
            F2, G2 = eng.eval(x)
            foms, grads = eng.member_results()
            xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda")
            fg = torch.zeros(K * N + 1, dtype=torch.float64, device="cuda")
            eng.eval_device(xd.data_ptr(), fg.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            h = fg.cpu().numpy()
        assert F == F2 and np.array_equal(G, G2)
        assert h[-1] == F and np.array_equal(h[:-1].reshape(N, K).T, G)
        assert F == foms[0] * wts[0] and np.array_equal(G, grads[0] * wts[0])
        assert_parity(F, G, F_ref, G_ref, n, what=f"{flow}, direct publication {direct}")
        res.append((F, G))
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1])


def test_flow_by_ensemble_size(qoc, oracle, monkeypatch):
    """9 <= n <= 16, rank-one states: chunked sweep_thin.hip chain for a handful of members, the propagator chain on DPP
    products from 80, the Taylor flow from 15/16 x compute units (lowered here)."""
    for name in ("GRAPE_ACTION", "GRAPE_THIN_DPP", "GRAPE_HOIST"):
        monkeypatch.delenv(name, raising=False)
    monkeypatch.setenv("GRAPE_ACTION_MIN", "120")
    n, K, N = 16, 2, 40
    seen = {}
    for E in (4, 90, 130):
        A, B, Xi, Xt, wts, x = _problem(n, K, N, E, True, False, True, seed=E)
        with qoc.GrapeEngine("CoherenceTransfer", A, B, Xi, Xt, wts, 1.0, N, member_results=True) as eng:
            info = eng.info
            seen[E] = (info["expm_action"], info["time_chunks"] >= 2)
            F, G = eng.eval(x)
            foms, grads = eng.member_results()
        for k in (0, E - 1):
            f_ref, g_ref = oracle.member_eval("CoherenceTransfer", A[k], B[k], Xi[k], Xt[k], x, 1.0)[:2]
            assert_parity(foms[k], grads[k], f_ref, g_ref, n, what=f"E = {E}, member {k}")
    assert seen == {4: (0, True), 90: (0, False), 130: (1, False)}
