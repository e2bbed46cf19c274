"""GPU: the MFMA tile-kernel family (operator dimension 5..32, zero-padded to 16 or 32) against
the oracle: the configs BASELINE.json names C4 (16x16 Liouvillian, CoherenceTransfer, non-Hermitian
generator) and C5 (32x32 UnitaryGate) at parity-test sizes, plus odd sizes that exercise padding."""
import numpy as np
import pytest

from conftest import assert_parity

pytestmark = pytest.mark.gpu


def _engine(qoc, w, **kw):
    return qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, member_results=True, **kw)


def _random_problem(qoc, n, K, N, E, sys_type, seed, hermitian=True, mixed=False):
    """mixed: full-rank density operators (pure states are rank one and run the matrix-vector chain for 9 <= n <= 16,
    tests/test_gpu_thin.py; the dense chains need states that are not)."""
    rng = np.random.default_rng(seed)

    def rnd():
        M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        return (M + M.conj().T) / 2 if hermitian else M
    A = np.array([rnd() for _ in range(E)]) * 0.7
    B = np.array([[rnd() for _ in range(K)] for _ in range(E)]) * 0.5
    if sys_type == "UnitaryGate":
        Xi = np.array([np.eye(n, dtype=complex)] * E)
        Xt = np.array([np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))[0] for _ in range(E)])
    else:
        def pure():
            v = rng.standard_normal(n) + 1j * rng.standard_normal(n)
            v /= np.linalg.norm(v)
            return np.outer(v, v.conj())

        def rho():
            return 0.6 * pure() + 0.3 * pure() + 0.1 * pure() if mixed else pure()
        Xi = np.array([rho() for _ in range(E)])
        Xt = np.array([rho() for _ in range(E)])
    w = qoc.workloads.Workload("rnd", sys_type, n, K, N, E, 1.0, A, B, Xi, Xt, np.full(E, 1.0 / E),
                               rng.uniform(0, 1, (K, N)))
    return w


@pytest.mark.parametrize("n,sys_type,herm", [(5, "UnitaryGate", True), (8, "StateTransfer", True),
                                             (16, "UnitaryGate", True), (16, "CoherenceTransfer", False),
                                             (17, "StateTransfer", True), (27, "UnitaryGate", False),
                                             (32, "UnitaryGate", True), (32, "StateTransfer", True)])
@pytest.mark.parametrize("variant", [0, 1])
def test_tile_family_random(qoc, oracle, n, sys_type, herm, variant):
    w = _random_problem(qoc, n, 3, 12, 3, sys_type, seed=100 + n, hermitian=herm)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T,
                                                            variant=variant, per_member=True)
    with _engine(qoc, w, variant=variant, flags=qoc.engine.FLAG_KEEP_COSTATES) as eng:
        F, G = eng.eval(w.x)
        foms, grads = eng.member_results()
        assert eng.info["kernel_family"] == 1
        P, X, L = eng.trajectory(1, costates=True)
    _, _, Pr, Xr, Lr = oracle.member_eval(w.sys_type, w.A[1], w.B[1], w.Xi[1], w.Xt[1], w.x, w.T, variant=variant,
                                          trajectory=True)
    for got, want, what in ((P, Pr, "propagators"), (X, Xr, "states"), (L, Lr, "costates")):
        assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), what
    for k in range(w.E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], w.n, what=f"n={n} member {k}")
    assert_parity(F, G, F_ref, G_ref, w.n, what=f"n={n}")


@pytest.mark.parametrize("n,sys_type", [(6, "UnitaryGate"), (8, "StateTransfer"), (16, "UnitaryGate"),
                                        (16, "StateTransfer"), (23, "StateTransfer"), (32, "UnitaryGate"),
                                        (32, "StateTransfer")])
@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("chain", ["chunked", "sequential"])
def test_tile_family_unitary_flow(qoc, oracle, monkeypatch, n, sys_type, variant, chain):
    """Hermitian generators: the chain kernel carries M_t = P' M P and stores no forward states.  Small ensembles cut
    the time axis into chunks evaluated in parallel (grape_info.time_chunks); GRAPE_NO_TP=1 keeps the one-wavefront
    chain large ensembles run."""
    if chain == "sequential":
        monkeypatch.setenv("GRAPE_NO_TP", "1")
    w = _random_problem(qoc, n, 3, 14, 3, sys_type, seed=300 + n, hermitian=True, mixed=True)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T,
                                                            variant=variant, per_member=True)
    with _engine(qoc, w, variant=variant) as eng:
        F, G = eng.eval(w.x)
        foms, grads = eng.member_results()
        assert eng.info["kernel_family"] == 1 and eng.info["unitary_flow"] == 1 and eng.info["rank_one_chain"] == 0
        assert (eng.info["time_chunks"] >= 2) == (chain == "chunked")
        with pytest.raises(qoc.GrapeError):
            eng.trajectory(0)                               # no stored states in this flow
    for k in range(w.E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], w.n, what=f"n={n} member {k}")
    assert_parity(F, G, F_ref, G_ref, w.n, what=f"n={n}")


@pytest.mark.parametrize("mixed", [True, False])
def test_many_controls(qoc, oracle, mixed):
    """K = 17 controls at n = 12: neither the generator tiles nor the transposed operators fit
    their LDS caches, so the kernels take their global-memory paths (dense chain, and the rank-one chain)."""
    w = _random_problem(qoc, 12, 17, 9, 2, "StateTransfer", seed=5, mixed=mixed)
    F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T)
    with _engine(qoc, w) as eng:
        assert eng.info["rank_one_chain"] == (0 if mixed else 1)
        F, G = eng.eval(w.x)
    assert_parity(F, G, F_ref, G_ref, w.n, what="K=17")


@pytest.mark.parametrize("dense", [False, True])
def test_c4_liouvillian_parity(qoc, oracle, dense):
    """BASELINE config 4 at parity size: 16x16 Liouvillian superoperators, CoherenceTransfer, K=4, N=1000.  Its states
    vec(rho) vec(rho)' are rank one: the default is the matrix-vector chain; dense = the MFMA chain on the same inputs."""
    w = qoc.workloads.config("C4", E=6)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T,
                                                            per_member=True)
    with _engine(qoc, w, flags=qoc.engine.FLAG_FORCE_GENERAL if dense else 0) as eng:
        assert eng.info["rank_one_chain"] == (0 if dense else 1)
        F, G = eng.eval(w.x)
        foms, grads = eng.member_results()
    for k in range(w.E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], w.n, what=f"C4 member {k}")
    assert_parity(F, G, F_ref, G_ref, w.n, what="C4")


@pytest.mark.parametrize("sys_type,n,K,N,E,mixed", [("CoherenceTransfer", 16, 4, 41, 3, False), ("StateTransfer", 12, 3, 7, 2, True),
                                                    ("UnitaryGate", 9, 2, 6, 2, False), ("StateTransfer", 16, 17, 9, 2, True),
                                                    ("StateTransfer", 16, 4, 4, 2, False), ("CoherenceTransfer", 16, 4, 64, 2, False)])
def test_two_wave_chain_meets_anywhere_on_the_time_axis(qoc, oracle, monkeypatch, sys_type, n, K, N, E, mixed):
    """chain_tile_split_kernel (wave 0 forward from t = 0, wave 1 backward from t = N, meeting at N / 2): sandwich and
    left-multiplication flows, the three Hermitian specialisations (mixed / pure density operators with Hermitian controls,
    general states), list traces in registers (K = 4 x 64 entries), in LDS and dense operators (K beyond the LDS cache), slice
    counts that are / are not multiples of the prefetch rings -- and the meeting point moved to both ends and off-centre
    (GRAPE_TILE_SPLIT_PERMILLE): every placement gives the oracle's numbers."""
    monkeypatch.setenv("GRAPE_NO_TP", "1")
    if sys_type == "CoherenceTransfer":
        w = qoc.workloads.config("C4", E=E, N=N)
    else:
        w = _random_problem(qoc, n, K, N, E, sys_type, seed=11, mixed=mixed)
        if K == 4:
            w.B[:] = np.triu(w.B) * (np.abs(w.B) > 0.45)          # few non-zeros, not Hermitian: HERM = 1 with lists in registers
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T, per_member=True)
    res = []
    for permille in (None, 0, 1000, 333, 800):
        if permille is None:
            monkeypatch.delenv("GRAPE_TILE_SPLIT_PERMILLE", raising=False)
        else:
            monkeypatch.setenv("GRAPE_TILE_SPLIT_PERMILLE", str(permille))
        with _engine(qoc, w, flags=qoc.engine.FLAG_FORCE_GENERAL) as eng:
            F, G = eng.eval(w.x)
            foms, grads = eng.member_results()
            assert any("chain_tile_split_kernel" in k for k in eng.kernel_names()), eng.kernel_names()
            assert eng.info["states_stored"] == 0
        for k in range(w.E):
            assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], w.n, what=f"meeting at {permille} permille, member {k}")
        assert_parity(F, G, F_ref, G_ref, w.n, what=f"meeting at {permille} permille")
        res.append(G)
    for G in res[1:]:
        assert np.max(np.abs(G - res[0])) <= 1e-10 * max(1.0, np.max(np.abs(res[0])))


def test_two_wave_chain_on_ensembles_beyond_the_resident_wave_slots(qoc, oracle, monkeypatch):
    """2304 full-rank 16 x 16 members (more workgroups than the device holds at once: they queue): every member against the oracle"""
    monkeypatch.setenv("GRAPE_NO_THIN", "1")
    w = qoc.workloads.config("C4", E=2304, N=8)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T, per_member=True)
    with _engine(qoc, w) as eng:
        F, G = eng.eval(w.x)
        foms, grads = eng.member_results()
        assert any("chain_tile_split_kernel" in k for k in eng.kernel_names()), eng.kernel_names()
    assert_parity(F, G, F_ref, G_ref, w.n, what="E = 2304")
    worst = max(np.abs(grads[k] - grads_ref[k]).max() / max(np.abs(grads_ref[k]).max(), 1e-30) for k in range(w.E))
    assert worst <= 1e-10 and np.abs(foms - foms_ref).max() <= 1e-10, (worst, np.abs(foms - foms_ref).max())


def test_two_wave_chain_forms_its_propagators_when_the_controls_are_shared(qoc, oracle, monkeypatch):
    """member-invariant control operators: chain_tile_split_kernel<.., EXPM> builds P_t in its phase 1 from the pre-pass's
    control sums (no expm kernel in the launch list) and still hands out the propagators; GRAPE_SPLIT_EXPM=0 keeps
    prop_hoist1_kernel + the chain that reads P_t -- same numbers; members with their own controls never take the fused form.
    Large generators (squarings) included."""
    monkeypatch.setenv("GRAPE_NO_TP", "1")
    monkeypatch.setenv("GRAPE_NO_THIN", "1")
    for scale, N in ((1.0, 21), (9.0, 6)):
        w = qoc.workloads.config("C4", E=9, N=N)                        # (the control sum is hoisted from 8 members on)
        w.A = w.A * scale
        F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T, per_member=True)
        res = {}
        for fused in (True, False):
            if fused:
                monkeypatch.delenv("GRAPE_SPLIT_EXPM", raising=False)
            else:
                monkeypatch.setenv("GRAPE_SPLIT_EXPM", "0")
            with _engine(qoc, w, max_batch=2) as eng:
                F, G = eng.eval(w.x)
                foms, grads = eng.member_results()
                names = eng.kernel_names()
                P = eng.trajectory(w.E - 1, states=False)[0]
                Fb, Gb = eng.eval_batch(np.array([0.3 * w.x, w.x]))
            assert any("chain_tile_split_kernel" in k for k in names) and ("prop_hoist1_kernel" in names) == (not fused), names
            assert "ctrl_sum_kernel" in names
            for k in range(w.E):
                assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], w.n, what=f"fused={fused} scale={scale} member {k}")
            assert_parity(F, G, F_ref, G_ref, w.n, what=f"fused={fused} scale={scale}")
            assert Fb[1] == F and np.array_equal(Gb[1], G)
            P_ref = oracle.member_eval(w.sys_type, w.A[-1], w.B[-1], w.Xi[-1], w.Xt[-1], w.x, w.T, trajectory=True)[2]
            assert np.abs(P - P_ref).max() <= 1e-12 * max(1.0, np.abs(P_ref).max())
            res[fused] = (F, G)
        assert abs(res[True][0] - res[False][0]) <= 1e-12 and np.abs(res[True][1] - res[False][1]).max() <= 1e-12 * max(1.0, np.abs(G_ref).max())
    w = qoc.workloads.config("C4", E=3, N=12)
    w.B = w.B * np.array([1.0, 1.1, 0.9])[:, None, None, None]         # amplitude-scaled controls per member
    monkeypatch.delenv("GRAPE_SPLIT_EXPM", raising=False)
    with _engine(qoc, w) as eng:
        F, G = eng.eval(w.x)
        names = eng.kernel_names()
    assert any(k.startswith("prop_") for k in names) and "ctrl_sum_kernel" not in names, names
    F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T)
    assert_parity(F, G, F_ref, G_ref, w.n, what="per-member controls")


def test_many_controls_keep_the_one_wave_chain(qoc, oracle, monkeypatch):
    """K = 70 dense control operators on 12 x 12 mixed states: beyond what the two-wave chain stores per slice in one instruction
    (K <= 64), the general flow stays with chain_tile_kernel -- same numbers"""
    monkeypatch.setenv("GRAPE_NO_TP", "1")
    w = _random_problem(qoc, 12, 70, 6, 2, "StateTransfer", seed=21, hermitian=False, mixed=True)
    w.B *= 0.1
    F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T)
    with _engine(qoc, w) as eng:
        F, G = eng.eval(w.x)
        names = eng.kernel_names()
    assert any("chain_tile_kernel" in k for k in names) and not any("chain_tile_split_kernel" in k for k in names), names
    assert_parity(F, G, F_ref, G_ref, w.n, what="K = 70")


@pytest.mark.parametrize("chain", ["chunked", "sequential"])
def test_c5_five_qubit_parity(qoc, oracle, monkeypatch, chain):
    """BASELINE config 5 at parity size: 32x32 UnitaryGate, K=6, N=2000 (2 members): the chunked time axis small
    ensembles get, and the one-wavefront chain of the full-size ensemble."""
    if chain == "sequential":
        monkeypatch.setenv("GRAPE_NO_TP", "1")
    w = qoc.workloads.config("C5", E=2)
    F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T)
    with _engine(qoc, w) as eng:
        assert (eng.info["time_chunks"] >= 2) == (chain == "chunked")
        F, G = eng.eval(w.x)
    assert_parity(F, G, F_ref, G_ref, w.n, what="C5")


@pytest.mark.parametrize("n,sys_type,E,N,chunks", [(7, "UnitaryGate", 5, 37, 5), (8, "StateTransfer", 1, 64, 9),
                                                   (16, "UnitaryGate", 1, 101, 13), (16, "StateTransfer", 4, 50, 2),
                                                   (24, "UnitaryGate", 2, 33, 16), (32, "StateTransfer", 1, 41, 4)])
def test_time_chunks_ragged_and_entry_points(qoc, oracle, monkeypatch, n, sys_type, E, N, chunks):
    """The chunked time axis with chunk counts that do not divide N (ragged last chunk), pair-packed members (n <= 8),
    single problems, and through the batched and device entry points."""
    import torch
    monkeypatch.setenv("GRAPE_TP_CHUNKS", str(chunks))
    w = _random_problem(qoc, n, 3, N, E, sys_type, seed=77 + n + N, hermitian=True, mixed=True)
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, member_results=True, max_batch=3) as eng:
        S = -(-N // chunks)
        assert eng.info["time_chunks"] == -(-N // S) and eng.info["unitary_flow"] == 1
        F, G = eng.eval(w.x)
        foms, grads = eng.member_results()
        xs = np.stack([w.x, -0.5 * w.x, 0.25 * w.x + 0.1])
        Fb, Gb = eng.eval_batch(xs)
        x_dev = torch.as_tensor(np.ascontiguousarray(w.x.T), device="cuda")
        fg = torch.zeros(w.K * w.N + 1, dtype=torch.float64, device="cuda")
        eng.eval_device(x_dev.data_ptr(), fg.data_ptr())
        torch.cuda.synchronize()
        fg = fg.cpu().numpy()
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T, per_member=True)
    for k in range(E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"member {k}")
    assert_parity(F, G, F_ref, G_ref, n, what="ensemble")
    assert_parity(fg[-1], fg[:-1].reshape(w.N, w.K).T, F_ref, G_ref, n, what="device entry point")
    for b in range(3):
        Fr, Gr = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, xs[b], w.T)
        assert_parity(Fb[b], Gb[b], Fr, Gr, n, what=f"batch entry {b}")


@pytest.mark.parametrize("name,E,members,dense", [("C4", 1024, (0, 1, 511, 1023), False), ("C4", 1024, (0, 1023), True),
                                                  ("C4", 1024, (0, 510, 1023), "expm"),
                                                  ("C5", 1024, (0, 1023), False), ("C5", 4096, (0, 2047, 4095), False),
                                                  ("C6", 256, (0, 127, 255), False),          # 64 x 64, N = 500: the bench's C6 line
                                                  ("C4", 1024, (0, 700, 1023), "pm"), ("C5", 1024, (0, 1023), "pm")])
def test_full_size_spot_members(qoc, oracle, name, E, members, dense, monkeypatch):
    """Full BASELINE sizes (C4: E = 1024, N = 1000 -- the vector flow the bench runs, the two-wave dense chain, and the
    MFMA expm + fused vector chain every ensemble with more than six per-member controls takes (GRAPE_ACTION=0); C5:
    E = 1024 of 4096, N = 2000, the unitary tile flow, and the whole E = 4096 ensemble on ONE GPU as the bench times it --
    134 GB of propagators): spot members against the oracle, the weighted sum, reproducibility.
    The small-E parity tests above take different launch branches (E < 2048, pack2, LDS fit)."""
    w = qoc.workloads.config(name, E=E)
    if dense == "expm":
        monkeypatch.setenv("GRAPE_ACTION", "0")
    if dense == "pm":                                      # bench.py's C4pm / C5pm: B_k = (1 + eps_k) B, the hoisted flows x s_k
        w.B = np.ascontiguousarray(w.B * (1.0 + 0.05 * (np.arange(w.E) / w.E - 0.5))[:, None, None, None])
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N,
                         flags=qoc.engine.FLAG_FORCE_GENERAL if dense is True else 0) as eng:
        F, G = eng.eval(w.x)
        if dense == "expm":
            assert eng.info["expm_action"] == 0 and eng.info["rank_one_chain"] == 1 and eng.info["fused_forward"] == 1
        elif name == "C4" and (not dense or dense == "pm"):
            assert eng.info["expm_action"] == 1
        if dense == "pm":
            names = eng.kernel_names()
            assert ("action_rows_kernel" in names and "action_parts_kernel" in names) if name == "C4" else "ctrl_sum_kernel" in names, names
        foms, grads = eng.member_results()
        F2, G2 = eng.eval(w.x)
    assert F == F2 and np.array_equal(G, G2)
    assert abs(F - foms @ w.wts) <= 1e-12 * max(1.0, abs(F))
    assert np.abs(G - np.tensordot(w.wts, grads, 1)).max() <= 1e-13 * np.abs(G).max() + 1e-16
    for k in members:
        F_ref, g_ref = oracle.member_eval(w.sys_type, w.A[k], w.B[k], w.Xi[k], w.Xt[k], w.x, w.T)
        assert_parity(foms[k], grads[k], F_ref, g_ref, w.n, what=f"{name} member {k}")


def test_c4_survey_target_ensemble_metric(qoc, oracle):
    """SURVEY.md 8d's original C4 target (|00> -> |11>): far-detuned members have gradients ~1e-10 of O(1)
    terms, so only the norm-wise ENSEMBLE metric of the north star is applied (per-member parity is covered
    by the generic-target case above)."""
    w = qoc.workloads.config("C4", E=12, N=1000, target="survey")
    F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T, n_threads=8)
    for flags in (0, qoc.engine.FLAG_FORCE_GENERAL):
        with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, flags=flags) as eng:
            F, G = eng.eval(w.x)
        assert_parity(F, G, F_ref, G_ref, w.n, what=f"C4 survey target, ensemble (flags {flags})")


def _sparse_problem(qoc, n, K, N, E, sys_type, seed, nnz_pairs=12):
    """Hermitian generators whose CONTROL operators are sparse (a few symmetric off-diagonal pairs + diagonal entries),
    like the Pauli-type controls of the BASELINE configs; mixed states so that the dense state chains run."""
    w = _random_problem(qoc, n, K, N, E, sys_type, seed, hermitian=True, mixed=True)
    rng = np.random.default_rng(seed + 1)
    B = np.zeros_like(w.B)
    for k in range(E):
        for c in range(K):
            for _ in range(nnz_pairs):
                i, j = rng.integers(0, n, 2)
                v = rng.standard_normal() + 1j * rng.standard_normal()
                if i == j:
                    B[k, c, i, i] = v.real
                else:
                    B[k, c, i, j] = v
                    B[k, c, j, i] = np.conj(v)
    w.B = B
    return w


@pytest.mark.parametrize("n,sys_type,K", [(9, "UnitaryGate", 3), (16, "UnitaryGate", 16), (16, "StateTransfer", 4),
                                          (20, "UnitaryGate", 6), (32, "UnitaryGate", 6), (32, "StateTransfer", 2),
                                          (27, "StateTransfer", 5)])
@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("chain", ["chunked", "sequential"])
def test_sparse_control_operators(qoc, oracle, monkeypatch, n, sys_type, K, variant, chain):
    """control operators with <= 64 non-zeros: the unitary chain reads (coefficient, position) lists from LDS instead of
    dense transposed operators; same results as the oracle and as the dense path (GRAPE_NO_SPARSE=1)."""
    if chain == "sequential":
        monkeypatch.setenv("GRAPE_NO_TP", "1")
    w = _sparse_problem(qoc, n, K, 15, 3, sys_type, seed=40 + n + K)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T,
                                                            variant=variant, per_member=True)
    with _engine(qoc, w, variant=variant) as eng:
        assert eng.info["sparse_controls"] == 1 and eng.info["unitary_flow"] == 1 and eng.info["rank_one_chain"] == 0
        F, G = eng.eval(w.x)
        foms, grads = eng.member_results()
    for k in range(w.E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], w.n, what=f"n={n} member {k}")
    assert_parity(F, G, F_ref, G_ref, w.n, what=f"n={n}")
    monkeypatch.setenv("GRAPE_NO_SPARSE", "1")
    with _engine(qoc, w, variant=variant) as eng:
        assert eng.info["sparse_controls"] == 0
        F_d, G_d = eng.eval(w.x)
    assert_parity(F, G, F_d, G_d, w.n, what=f"n={n} sparse vs dense lists")


@pytest.mark.parametrize("n,sys_type,K,keep", [(12, "StateTransfer", 3, True), (16, "UnitaryGate", 9, True),
                                                 (24, "StateTransfer", 4, False), (32, "UnitaryGate", 6, False),
                                                 (32, "CoherenceTransfer", 5, True)])
@pytest.mark.parametrize("variant", [0, 1])
def test_sparse_control_operators_general_flow(qoc, oracle, monkeypatch, n, sys_type, K, keep, variant):
    """non-Hermitian drift (general flow, chain_tile_kernel) with sparse control operators; keep: the debug flow,
    which also runs that kernel for n <= 16 (small ensembles otherwise take the two-wave split kernel)."""
    w = _sparse_problem(qoc, n, K, 13, 3, sys_type, seed=70 + n + K)
    rng = np.random.default_rng(n)
    w.A = w.A + 0.2j * np.array([np.diag(rng.uniform(-1, 0, n)) for _ in range(w.E)])      # damping: not Hermitian
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T,
                                                            variant=variant, per_member=True)
    flags = qoc.engine.FLAG_KEEP_COSTATES if keep else 0
    with _engine(qoc, w, variant=variant, flags=flags) as eng:
        assert eng.info["sparse_controls"] == 1 and eng.info["unitary_flow"] == 0
        F, G = eng.eval(w.x)
        foms, grads = eng.member_results()
    for k in range(w.E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], w.n, what=f"n={n} member {k}")
    assert_parity(F, G, F_ref, G_ref, w.n, what=f"n={n}")
    monkeypatch.setenv("GRAPE_NO_SPARSE", "1")
    with _engine(qoc, w, variant=variant, flags=flags) as eng:
        F_d, G_d = eng.eval(w.x)
    assert_parity(F, G, F_d, G_d, w.n, what=f"n={n} sparse vs dense lists")


def test_c5_controls_are_sparse(qoc):
    w = qoc.workloads.config("C5", E=2)
    with _engine(qoc, w) as eng:
        assert eng.info["sparse_controls"] == 1


@pytest.mark.parametrize("n,sys_type,E,N,chunks", [(6, "StateTransfer", 4, 40, 0), (8, "UnitaryGate", 1, 64, 7),
                                                   (12, "CoherenceTransfer", 3, 37, 5), (16, "UnitaryGate", 1, 200, 0),
                                                   (16, "StateTransfer", 2, 100, 9), (20, "StateTransfer", 1, 33, 4),
                                                   (32, "UnitaryGate", 2, 48, 0)])
@pytest.mark.parametrize("chain", ["chunked", "sequential"])
def test_general_flow_time_chunks(qoc, oracle, monkeypatch, n, sys_type, E, N, chunks, chain):
    """Non-Hermitian generators with full-rank states (the general flow: forward states stored, costates pulled back):
    small ensembles cut the time axis into chunks between the prefix and suffix products of the chunk products;
    pair-packed members, ragged chunks, 32 x 32 tiles, and the sequential chains (GRAPE_NO_TP=1) on the same inputs."""
    if chain == "sequential":
        monkeypatch.setenv("GRAPE_NO_TP", "1")
    elif chunks:
        monkeypatch.setenv("GRAPE_TP_CHUNKS", str(chunks))
    w = _random_problem(qoc, n, 3, N, E, sys_type, seed=500 + n + N, hermitian=False, mixed=True)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T, per_member=True)
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, member_results=True, max_batch=2) as eng:
        info = eng.info
        assert info["unitary_flow"] == 0 and info["rank_one_chain"] == 0 and (info["time_chunks"] >= 2) == (chain == "chunked")
        F, G = eng.eval(w.x)
        foms, grads = eng.member_results()
        xs = np.stack([w.x, 0.2 - 0.5 * w.x])
        Fb, Gb = eng.eval_batch(xs)
    for k in range(E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"member {k}")
    assert_parity(F, G, F_ref, G_ref, n, what="ensemble")
    Fr, Gr = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, xs[1], w.T)
    assert_parity(Fb[1], Gb[1], Fr, Gr, n, what="batch entry 1")


@pytest.mark.parametrize("n,K,N,sys_type,pauli", [
    (5, 2, 6, "UnitaryGate", False), (8, 3, 100, "StateTransfer", False), (16, 4, 5, "UnitaryGate", True),
    (16, 3, 300, "UnitaryGate", True), (16, 2, 129, "StateTransfer", False), (17, 2, 40, "UnitaryGate", False),
    (32, 6, 7, "UnitaryGate", True), (32, 5, 400, "UnitaryGate", True), (32, 2, 90, "StateTransfer", False)])
def test_single_unitary_problem_is_closed_by_the_chain_kernel(qoc, oracle, monkeypatch, n, K, N, sys_type, pauli):
    """ONE problem with Hermitian generators (the single-`Problem` closure, src/solve.jl:63-143), n = 5..32: the unitary
    chain kernel (one wave or four per product, chunked time axis or not, dense or list traces, member pairs for n <= 8)
    writes the weighted [G, F] itself and its last workgroup publishes it -- bitwise what the reduce kernel hands out
    (GRAPE_DIRECT_PUBLISH=0), on the host path and through the device entry point."""
    import torch
    w = _random_problem(qoc, n, K, N, 1, sys_type, seed=5 * n + N, mixed=True)
    if pauli:                                                # sparse control operators: the (coefficient, position) lists
        nq = {16: 4, 32: 5}[n]
        rng = np.random.default_rng(n + K)
        P = [np.eye(2), np.array([[0, 1], [1, 0]]), np.array([[0, -1j], [1j, 0]]), np.array([[1, 0], [0, -1]])]

        def string():
            M = np.array([[1.0 + 0j]])
            for _ in range(nq):
                M = np.kron(M, P[int(rng.integers(0, 4))])
            return M
        w.B[:] = np.array([[0.5 * string() for _ in range(K)]])
    w.wts[:] = 0.43
    F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T)
    res = []
    for direct in ("1", "0"):
        monkeypatch.setenv("GRAPE_DIRECT_PUBLISH", direct)
        with _engine(qoc, w) as eng:
            assert eng.info["unitary_flow"] == 1 and eng.info["kernel_family"] == 1
            F, G = eng.eval(w.x)
            F2, G2 = eng.eval(w.x)
            foms, grads = eng.member_results()
            xd = torch.as_tensor(np.ascontiguousarray(w.x.T), device="cuda")
            fg = torch.zeros(K * N + 1, dtype=torch.float64, device="cuda")
            eng.eval_device(xd.data_ptr(), fg.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            h = fg.cpu().numpy()
            F3, G3 = eng.eval(0.5 * w.x)
        assert F == F2 and np.array_equal(G, G2)
        assert h[-1] == F and np.array_equal(h[:-1].reshape(N, K).T, G)
        assert F == foms[0] * w.wts[0] and np.array_equal(G, grads[0] * w.wts[0])
        assert_parity(F, G, F_ref, G_ref, n, what=f"direct publication {direct}")
        res.append((F, G, F3, G3))
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1])
    assert res[0][2] == res[1][2] and np.array_equal(res[0][3], res[1][3])


@pytest.mark.parametrize("n,sys_type,K,E,N,pairs,herm", [
    (16, "UnitaryGate", 3, 3, 15, 40, True), (16, "StateTransfer", 6, 2, 70, 100, True), (32, "UnitaryGate", 6, 1, 120, 80, True),
    (32, "UnitaryGate", 4, 3, 9, 120, True), (32, "StateTransfer", 2, 2, 33, 128, True), (24, "UnitaryGate", 5, 4, 21, 60, True),
    (16, "CoherenceTransfer", 4, 3, 40, 90, False), (32, "CoherenceTransfer", 3, 2, 13, 100, False),
    (12, "StateTransfer", 2, 5, 64, 60, False)])
def test_control_lists_longer_than_a_wavefront(qoc, oracle, monkeypatch, n, sys_type, K, E, N, pairs, herm):
    """Control operators with 65 .. 256 non-zeros (sums of a few Pauli strings, global drives): lists of 128 / 192 / 256
    entries, a lane owns several of them -- unitary chain (one wave and four per product), general flow, split chain;
    against the oracle, the dense traces (GRAPE_NO_SPARSE=1), and the 64-entry rule of rounds 1-2 (GRAPE_SPARSE_MAX=64)."""
    w = _sparse_problem(qoc, n, K, N, E, sys_type, seed=11 + n + K + N, nnz_pairs=pairs)
    nz = max(int(np.count_nonzero(w.B[k, c])) for k in range(E) for c in range(K))
    assert 64 < nz <= 256, nz
    if not herm:
        rng = np.random.default_rng(n)
        w.A = w.A + 0.2j * np.array([np.diag(rng.uniform(-1, 0, n)) for _ in range(E)])
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T, per_member=True)
    with _engine(qoc, w) as eng:
        assert eng.info["sparse_controls"] == 1 and eng.info["unitary_flow"] == (1 if herm else 0)
        F, G = eng.eval(w.x)
        F2, G2 = eng.eval(w.x)
        foms, grads = eng.member_results()
    assert F == F2 and np.array_equal(G, G2)
    for k in range(E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"member {k}")
    assert_parity(F, G, F_ref, G_ref, n, what="long lists")
    monkeypatch.setenv("GRAPE_SPARSE_MAX", "64")
    with _engine(qoc, w) as eng:
        assert eng.info["sparse_controls"] == 0
        F_d, G_d = eng.eval(w.x)
    assert_parity(F, G, F_d, G_d, n, what="long lists vs dense traces")


def test_too_many_list_entries_keep_the_dense_traces(qoc):
    """K x list length beyond the kernels' LDS budget (1536 entries), or an operator with more than 256 non-zeros."""
    for n, K, pairs in ((32, 8, 128), (32, 2, 200)):
        w = _sparse_problem(qoc, n, K, 8, 2, "UnitaryGate", seed=3, nnz_pairs=pairs)
        with _engine(qoc, w) as eng:
            assert eng.info["sparse_controls"] == 0


@pytest.mark.parametrize("n,sys_type,E,N", [(16, "UnitaryGate", 1100, 64), (8, "StateTransfer", 2300, 40), (12, "UnitaryGate", 3000, 33)])
def test_single_tile_chains_are_chunked_beyond_one_wave_per_simd(qoc, oracle, n, sys_type, E, N):
    """n <= 16, Hermitian generators, full-rank states, MORE members than SIMDs: the chains (four waves fit a SIMD) still
    cut the time axis (round 3: up to 16 x CUs units) -- every member and the ensemble against the oracle."""
    w = _random_problem(qoc, n, 3, N, E, sys_type, seed=n + N, mixed=True)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T, per_member=True)
    with _engine(qoc, w) as eng:
        info = eng.info
        assert info["unitary_flow"] == 1 and info["time_chunks"] >= 2, info
        F, G = eng.eval(w.x)
        F2, G2 = eng.eval(w.x)
        foms, grads = eng.member_results()
    assert F == F2 and np.array_equal(G, G2)
    for k in list(range(0, E, 97)) + [E - 1]:
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"member {k}")
    assert_parity(F, G, F_ref, G_ref, n, what="ensemble")
