"""GPU: the matrix-vector chain for rank-one states in the 9 <= n <= 16 family (csrc/sweep_thin.hip) --
sandwich problems with Xi = v v', Xt = w w' and left multiplication of n x 1 states -- against the oracle's DENSE
evaluation of the same inputs (the reference's formulas, src/GRAPE.jl:216-303) at the 1e-10 bar, and against the
library's own dense chain (GRAPE_FLAG_FORCE_GENERAL)."""
import numpy as np
import pytest

from conftest import assert_parity

pytestmark = pytest.mark.gpu


def _problem(n, K, N, E, sand, herm_gen, herm_ctrl, seed, mixed=False):
    rng = np.random.default_rng(seed)

    def gen(h):
        M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        return (M + M.conj().T) / 2 if h else M
    A = np.array([gen(herm_gen) for _ in range(E)]) * 0.6
    B = np.array([[gen(herm_ctrl) for _ in range(K)] for _ in range(E)]) * 0.4

    def vec():
        v = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        return v / np.linalg.norm(v)
    if sand:
        def rho():
            if mixed:
                return sum(p * np.outer(v, v.conj()) for p, v in zip((0.6, 0.3, 0.1), (vec(), vec(), vec())))
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
    (16, 4, 101, 2, "UnitaryGate", False, True),
]


@pytest.mark.parametrize("n,K,N,E,sys_type,herm_gen,herm_ctrl", CASES)
@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("fused", [False, True])
def test_rank_one_chain_matches_dense_oracle(qoc, oracle, monkeypatch, n, K, N, E, sys_type, herm_gen, herm_ctrl, variant,
                                             fused):
    # fused: the expm kernel walks a member's slices in order and runs the forward vector pass itself (chosen by the
    # library when one workgroup per member fills the device; forced here for the small shapes)
    monkeypatch.setenv("GRAPE_FORCE_FUSE" if fused else "GRAPE_NO_FUSE", "1")
    sand = sys_type != "UnitaryGate"
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, sand, herm_gen, herm_ctrl, seed=7 * n + N + K)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, 1.2, variant=variant,
                                                             per_member=True)
    with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.2, N, variant=variant, member_results=True) as eng:
        assert eng.info["rank_one_chain"] == 1 and eng.info["kernel_family"] == 1 and eng.info["states_stored"] == 0
        F, G = eng.eval(x)
        foms, grads = eng.member_results()
        P = eng.trajectory(E - 1, states=False)[0]
    for k in range(E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"member {k}")
    assert_parity(F, G, F_ref, G_ref, n, what="ensemble")
    # the propagators handed out are P_t for every t, although odd slices live transposed in the workspace
    if sand:
        P_ref = oracle.member_eval(sys_type, A[-1], B[-1], Xi[-1], Xt[-1], x, 1.2, variant=variant, trajectory=True)[2]
    else:
        P_ref = oracle.member_eval_rect(A[-1], B[-1], Xi[-1], Xt[-1], x, 1.2, variant=variant, trajectory=True)[2]
    assert np.abs(P - P_ref).max() <= 1e-12 * max(1.0, np.abs(P_ref).max())
    # the library's own dense chain on the same inputs
    with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.2, N, variant=variant, flags=qoc.engine.FLAG_FORCE_GENERAL) as eng:
        assert eng.info["rank_one_chain"] == 0
        F_d, G_d = eng.eval(x)
    assert_parity(F, G, F_d, G_d, n, what="thin vs dense chain")


@pytest.mark.parametrize("which", ["mixed", "nonhermitian", "scaled"])
def test_states_that_are_not_rank_one_keep_the_dense_chain(qoc, oracle, which):
    n, K, N, E = 16, 3, 20, 3
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, True, False, True, seed=5, mixed=(which == "mixed"))
    if which == "nonhermitian":
        Xt = Xt.copy()
        Xt[1, 2, 5] += 0.25                       # one member's target is no longer v v'
    if which == "scaled":
        Xi = Xi * 1.0                             # still rank one: c v v' with c > 0 is (sqrt(c) v)(sqrt(c) v)'
        Xi[0] *= 2.5
    F_ref, G_ref = oracle.ensemble_eval("CoherenceTransfer", A, B, Xi, Xt, wts, x, 0.9)
    with qoc.GrapeEngine("CoherenceTransfer", A, B, Xi, Xt, wts, 0.9, N) as eng:
        assert eng.info["rank_one_chain"] == (1 if which == "scaled" else 0)
        F, G = eng.eval(x)
    assert_parity(F, G, F_ref, G_ref, n, what=which)


def test_operators_can_switch_between_the_chains(qoc, oracle):
    """grape_set_operators decides per upload: rank one -> vectors (small record buffer), then mixed states ->
    dense chain (state dumps), and back, on the same context."""
    n, K, N, E = 16, 2, 30, 2
    A, B, Xi, Xt, wts, _ = _problem(n, K, N, E, True, False, True, seed=1)
    with qoc.GrapeEngine("CoherenceTransfer", A, B, Xi, Xt, wts, 1.0, N) as eng:
        for mixed in (False, True, False):
            A, B, Xi, Xt, wts, x = _problem(n, K, N, E, True, False, True, seed=11, mixed=mixed)
            eng.set_operators(A, B, Xi, Xt, wts)
            assert eng.info["rank_one_chain"] == (0 if mixed else 1)
            F, G = eng.eval(x)
            F_ref, G_ref = oracle.ensemble_eval("CoherenceTransfer", A, B, Xi, Xt, wts, x, 1.0)
            assert_parity(F, G, F_ref, G_ref, n, what=f"mixed={mixed}")


def test_batched_and_device_entry_points(qoc, oracle):
    import torch
    n, K, N, E = 16, 4, 50, 3
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, True, False, True, seed=3)
    rng = np.random.default_rng(0)
    xs = np.stack([x, rng.uniform(-1, 1, (K, N)), 0.5 * x])
    with qoc.GrapeEngine("CoherenceTransfer", A, B, Xi, Xt, wts, 1.0, N, max_batch=3) as eng:
        assert eng.info["rank_one_chain"] == 1
        Fs, Gs = eng.eval_batch(xs)
        xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda")
        fg = torch.zeros(K * N + 1, dtype=torch.float64, device="cuda")
        eng.eval_device(xd.data_ptr(), fg.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        h = fg.cpu().numpy()
    for b in range(3):
        F_ref, G_ref = oracle.ensemble_eval("CoherenceTransfer", A, B, Xi, Xt, wts, xs[b], 1.0)
        assert_parity(Fs[b], Gs[b], F_ref, G_ref, n, what=f"batch entry {b}")
    assert h[-1] == Fs[0] and np.array_equal(h[:-1].reshape(N, K).T, Gs[0])


def _random_cases(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        sys_type = str(rng.choice(["StateTransfer", "CoherenceTransfer", "UnitaryGate"]))
        out.append((i, int(rng.integers(9, 17)), int(rng.integers(1, 10)), int(rng.choice([1, 2, 3, 5, 6, 9, 31, 32, 33, 64, 127, 200])),
                    int(rng.choice([1, 2, 3, 7])), sys_type, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)),
                    int(rng.integers(0, 2))))
    return out


@pytest.mark.parametrize("i,n,K,N,E,sys_type,herm_gen,herm_ctrl,variant", _random_cases(40, 2027))
def test_rank_one_chain_random_shapes(qoc, oracle, i, n, K, N, E, sys_type, herm_gen, herm_ctrl, variant):
    sand = sys_type != "UnitaryGate"
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, sand, herm_gen, herm_ctrl, seed=900 + i)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, 0.8, variant=variant,
                                                             per_member=True)
    with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 0.8, N, variant=variant, member_results=True) as eng:
        assert eng.info["rank_one_chain"] == 1
        F, G = eng.eval(x)
        foms, grads = eng.member_results()
    for k in range(E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"case {i} member {k}")
    assert_parity(F, G, F_ref, G_ref, n, what=f"case {i}")


@pytest.mark.parametrize("sys_type", ["CoherenceTransfer", "UnitaryGate"])
@pytest.mark.parametrize("E", [1024, 520])
def test_fused_forward_pass_is_bitwise_the_separate_one(qoc, monkeypatch, sys_type, E):
    """From half a device of members on (2 x 256 CUs) the library fuses the forward vector pass into the expm kernel by
    itself.  With the round-2 expm kernel (GRAPE_HOIST=0) the same summation trees run in both flows and the results
    agree to the last bit; the round-3 kernel multiplies EVERY slice by its transposed registers (the short hand-over),
    a different tree on the even slices: agreement to rounding, each flow bitwise reproducible."""
    n, K, N = 16, 4, 37
    monkeypatch.setenv("GRAPE_ACTION", "0")          # (ensembles of this size would take the vector flow of action_thin.hip)
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, sys_type != "UnitaryGate", False, True, seed=5)
    for hoist in ("0", None):
        if hoist is None:
            monkeypatch.delenv("GRAPE_HOIST", raising=False)
        else:
            monkeypatch.setenv("GRAPE_HOIST", hoist)
        monkeypatch.delenv("GRAPE_NO_FUSE", raising=False)
        res = {}
        for mode in ("auto", "separate"):
            if mode == "separate":
                monkeypatch.setenv("GRAPE_NO_FUSE", "1")
            with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.2, N, member_results=True, max_batch=3) as eng:
                assert eng.info["rank_one_chain"] == 1 and eng.info["fused_forward"] == (1 if mode == "auto" else 0)
                F, G = eng.eval(x)
                F2, G2 = eng.eval(x)
                assert F == F2 and np.array_equal(G, G2)
                xs = np.stack([x, 0.5 * x, -x])
                Fb, Gb = eng.eval_batch(xs)
                res[mode] = (F, G.copy(), eng.member_results()[1].copy(), np.array(Fb), np.array(Gb))
        for a, b in zip(res["auto"], res["separate"]):
            a, b = np.asarray(a), np.asarray(b)
            if hoist == "0":
                assert np.array_equal(a, b)
            else:
                assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max())
        assert np.isfinite(res["auto"][1]).all() and abs(res["auto"][3][0] - res["auto"][0]) <= 1e-12 * max(1.0, abs(res["auto"][0]))


@pytest.mark.parametrize("n,K,N,E,sys_type,chunks", [(16, 4, 257, 1, "CoherenceTransfer", 0), (16, 3, 100, 3, "StateTransfer", 3),
                                                     (12, 2, 64, 2, "UnitaryGate", 8), (16, 4, 1000, 1, "UnitaryGate", 0),
                                                     (9, 5, 41, 4, "CoherenceTransfer", 2), (16, 6, 200, 2, "StateTransfer", 25)])
@pytest.mark.parametrize("chain", ["chunked", "sequential"])
def test_rank_one_chain_time_chunks(qoc, oracle, monkeypatch, n, K, N, E, sys_type, chunks, chain):
    """Small ensembles of rank-one problems cut the time axis into chunks of a multiple of 8 slices (dense chunk
    products, vector scan over the chunk boundaries, chunk-parallel vector sweeps: grape_info.time_chunks); ragged
    last chunks, the library's own chunk count (0) and forced ones, batched and device entry points -- and the
    one-wavefront chain (GRAPE_NO_TP=1) on the same inputs."""
    import torch
    monkeypatch.setenv("GRAPE_THIN_SINGLE", "1")      # (a single rank-one problem would take the dense chunked flows)
    if chain == "sequential":
        monkeypatch.setenv("GRAPE_NO_TP", "1")
    elif chunks:
        monkeypatch.setenv("GRAPE_TP_CHUNKS", str(chunks))
    sand = sys_type != "UnitaryGate"
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, sand, False, True, seed=11 * n + N)
    with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.2, N, member_results=True, max_batch=2) as eng:
        info = eng.info
        assert info["rank_one_chain"] == 1 and (info["time_chunks"] >= 2) == (chain == "chunked")
        if chain == "chunked" and chunks:
            S = -(-(-(-N // chunks)) // 8) * 8
            assert info["time_chunks"] == -(-N // S)
        F, G = eng.eval(x)
        foms, grads = eng.member_results()
        xs = np.stack([x, 0.3 - x])
        Fb, Gb = eng.eval_batch(xs)
        x_dev = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda")
        fg = torch.zeros(K * N + 1, dtype=torch.float64, device="cuda")
        eng.eval_device(x_dev.data_ptr(), fg.data_ptr())
        torch.cuda.synchronize()
        fg = fg.cpu().numpy()
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, 1.2, per_member=True)
    for k in range(E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"member {k}")
    assert_parity(F, G, F_ref, G_ref, n, what="ensemble")
    assert_parity(fg[-1], fg[:-1].reshape(N, K).T, F_ref, G_ref, n, what="device entry point")
    Fr, Gr = oracle.ensemble_eval(sys_type, A, B, Xi, Xt, wts, xs[1], 1.2)
    assert_parity(Fb[1], Gb[1], Fr, Gr, n, what="batch entry 1")


@pytest.mark.parametrize("sys_type,herm_gen", [("CoherenceTransfer", False), ("StateTransfer", True), ("UnitaryGate", False)])
def test_single_rank_one_problem(qoc, oracle, monkeypatch, sys_type, herm_gen):
    """ONE rank-one problem of at least 64 slices is latency-bound: the propagator chain of action_thin.hip on a chunked
    time axis (round 3); without it (GRAPE_DPP_CHUNKS=0) the dense flows with the chunked time axis, which are ahead of
    sweep_thin.hip's chunked chain there (general flow, or unitary flow for Hermitian generators)."""
    n, K, N = 16, 3, 96
    A, B, Xi, Xt, wts, x = _problem(n, K, N, 1, sys_type != "UnitaryGate", herm_gen, True, seed=3)
    F_ref, G_ref = oracle.ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, 1.2)
    for name in ("GRAPE_DPP_CHUNKS", "GRAPE_THIN_DPP", "GRAPE_HOIST", "GRAPE_ACTION"):
        monkeypatch.delenv(name, raising=False)
    with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.2, N) as eng:
        info = eng.info
        assert info["rank_one_chain"] == 1 and info["prop_chain"] == 1 and info["time_chunks"] >= 2 and info["unitary_flow"] == 0
        F, G = eng.eval(x)
    assert_parity(F, G, F_ref, G_ref, n, what="single rank-one problem, chunked propagator chain")
    monkeypatch.setenv("GRAPE_DPP_CHUNKS", "0")
    with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.2, N) as eng:
        info = eng.info
        assert info["rank_one_chain"] == 0 and info["time_chunks"] >= 2 and info["unitary_flow"] == (1 if herm_gen else 0)
        F, G = eng.eval(x)
    assert_parity(F, G, F_ref, G_ref, n, what="single rank-one problem, dense chunked flows")
