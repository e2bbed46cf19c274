"""GPU: operators beyond one wavefront's registers, n = 33 .. 64 (sweep_grid.hip: a workgroup of NT x NT waves per matrix,
NT = 3, 4; `_fom_and_gradient_GRAPE!` src/GRAPE.jl:25-96 is size-generic) against the oracle at the 1e-10 bar -- every
member's (F_k, g_k), the ensemble sums, and the stored propagators / states / costates; both formula variants, all three
system types, Hermitian and non-Hermitian generators, the squaring path of the expm, batches, n x m states, in-process
groups.  GRAPE_GRID=1 sends the small sizes through the same kernels (NT = 1, 2), where the tile family cross-checks them."""
import numpy as np
import pytest

from conftest import assert_parity
from test_gpu_tile import _engine, _random_problem, _sparse_problem

pytestmark = pytest.mark.gpu


def _check(qoc, oracle, w, variant=0, traj=True, **kw):
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T,
                                                            variant=variant, per_member=True)
    flags = qoc.engine.FLAG_KEEP_COSTATES if traj else 0
    with _engine(qoc, w, variant=variant, flags=flags, **kw) as eng:
        F, G = eng.eval(w.x)
        foms, grads = eng.member_results()
        info = eng.info
        names = eng.kernel_names()
        if traj:
            P, X, L = eng.trajectory(w.E - 1, costates=True)
    assert info["kernel_family"] == 1 and info["states_stored"] == 1
    assert any(k.startswith("grid_prop_kernel") for k in names) and any(k.startswith("grid_chain_kernel") for k in names), names
    if traj:
        k = w.E - 1
        _, _, Pr, Xr, Lr = oracle.member_eval(w.sys_type, w.A[k], w.B[k], w.Xi[k], w.Xt[k], w.x, w.T, variant=variant,
                                              trajectory=True)
        for got, want, what in ((P, Pr, "propagators"), (X, Xr, "states"), (L, Lr, "costates")):
            assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), what
    for k in range(w.E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], w.n, what=f"n={w.n} member {k}")
    assert_parity(F, G, F_ref, G_ref, w.n, what=f"n={w.n}")
    return F, G


@pytest.mark.parametrize("n", [33, 48, 64])
@pytest.mark.parametrize("sys_type,herm", [("UnitaryGate", True), ("UnitaryGate", False), ("StateTransfer", True),
                                           ("StateTransfer", False), ("CoherenceTransfer", False)])
@pytest.mark.parametrize("variant", [0, 1])
def test_grid_family_random(qoc, oracle, n, sys_type, herm, variant):
    w = _random_problem(qoc, n, 3, 10, 3, sys_type, seed=700 + n, hermitian=herm, mixed=True)
    # (norms as the BASELINE configs have them: dt |H| of a few hundredths .. tenths)
    w.A *= 0.25
    w.B *= 0.25
    _check(qoc, oracle, w, variant=variant)


@pytest.mark.parametrize("n", [40, 57])
def test_grid_family_odd_sizes_many_controls(qoc, oracle, n):
    """padding inside the last tile row / column; K beyond one trace group (4) and not a multiple of it"""
    w = _random_problem(qoc, n, 7, 6, 2, "StateTransfer", seed=811 + n, hermitian=True, mixed=True)
    w.A *= 0.2
    w.B *= 0.2
    _check(qoc, oracle, w)


@pytest.mark.parametrize("scale", [1.0, 6.0])
def test_grid_family_squaring_path(qoc, oracle, scale):
    """dt |H|_1 far above theta8 = 0.08: several squarings per slice"""
    w = _random_problem(qoc, 48, 2, 8, 2, "UnitaryGate", seed=99, hermitian=True)
    w.A *= scale
    w.B *= scale
    _check(qoc, oracle, w)


@pytest.mark.parametrize("n,sys_type,K,pairs", [(40, "UnitaryGate", 3, 12), (48, "StateTransfer", 11, 20), (64, "UnitaryGate", 6, 30),
                                                (64, "StateTransfer", 16, 12), (57, "UnitaryGate", 4, 100)])
@pytest.mark.parametrize("variant", [0, 1])
def test_grid_family_sparse_control_operators(qoc, oracle, monkeypatch, n, sys_type, K, pairs, variant):
    """control operators with few non-zeros (Pauli-type controls, as C6 has them): wave c of the workgroup forms control c's
    trace from its (coefficient, position) list and tr R from the image's diagonal -- lists of 64 .. 256 entries, more
    controls than waves (K = 11 on 9 waves), the hoisted H build where the members share them; same numbers as the oracle and
    as the dense traces (GRAPE_NO_SPARSE=1)."""
    w = _sparse_problem(qoc, n, K, 9, 3, sys_type, seed=60 + n + K, nnz_pairs=pairs)
    w.A *= 0.25
    w.B *= 0.25
    if K == 6:
        w.B[:] = w.B[0]                                   # member-invariant controls: ctrl_sum_kernel + grid_prop_kernel<., true>
    with _engine(qoc, w, variant=variant) as eng:
        F_sp, G_sp = eng.eval(w.x)
        assert eng.info["sparse_controls"] == 1 and eng.info["hoisted_controls"] == (1 if K == 6 else 0)
        names = eng.kernel_names()
    assert ("ctrl_sum_kernel" in names) == (K == 6)
    monkeypatch.setenv("GRAPE_NO_SPARSE", "1")
    F, G = _check(qoc, oracle, w, variant=variant)
    assert abs(F - F_sp) <= 1e-12 * max(1.0, abs(F)) and np.abs(G - G_sp).max() <= 1e-12 * max(1.0, np.abs(G).max())
    F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T, variant=variant)
    assert_parity(F_sp, G_sp, F_ref, G_ref, w.n, what=f"sparse traces n={n} K={K}")


def test_grid_family_single_problem_and_one_slice(qoc, oracle):
    w = _random_problem(qoc, 64, 2, 1, 1, "StateTransfer", seed=5, hermitian=True, mixed=True)
    w.A *= 0.2
    w.B *= 0.2
    _check(qoc, oracle, w)
    w = _random_problem(qoc, 33, 1, 5, 1, "UnitaryGate", seed=6, hermitian=False)
    w.A *= 0.2
    w.B *= 0.2
    _check(qoc, oracle, w, variant=1)


def test_grid_family_batches_and_reproducibility(qoc, oracle):
    w = _random_problem(qoc, 48, 3, 9, 4, "StateTransfer", seed=31, hermitian=True, mixed=True)
    w.A *= 0.2
    w.B *= 0.2
    rng = np.random.default_rng(3)
    xs = [w.x, rng.uniform(-1, 1, w.x.shape), rng.uniform(0, 2, w.x.shape)]
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, max_batch=3) as eng:
        Fb, Gb = eng.eval_batch(np.array(xs))
        singles = [eng.eval(x) for x in xs]
        again = eng.eval(xs[1])
    for b, x in enumerate(xs):
        F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, x, w.T)
        assert_parity(Fb[b], Gb[b], F_ref, G_ref, w.n, what=f"batch entry {b}")
        assert Fb[b] == singles[b][0] and np.array_equal(Gb[b], singles[b][1])          # a batch = its single evaluations, bitwise
    assert again[0] == singles[1][0] and np.array_equal(again[1], singles[1][1])         # run to run: bitwise


def test_grid_family_rectangular_states(qoc, oracle):
    """64 x 1 states (vec(rho) of a three-qubit system under a Liouvillian, test/liou.jl:38-48 one size up): zero-padded"""
    rng = np.random.default_rng(17)
    n, K, N, E = 64, 2, 6, 2
    A = np.array([0.15 * (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) for _ in range(E)])
    B = np.array([[0.1 * (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) for _ in range(K)]] * E)
    Xi = rng.standard_normal((E, n, 1)) + 1j * rng.standard_normal((E, n, 1))
    Xt = rng.standard_normal((E, n, 1)) + 1j * rng.standard_normal((E, n, 1))
    x = rng.uniform(0, 1, (K, N))
    wts = np.array([0.3, 0.7])
    F_ref, G_ref = oracle.ensemble_eval("UnitaryGate", A, B, Xi, Xt, wts, x, 1.0)
    with qoc.GrapeEngine("UnitaryGate", A, B, Xi, Xt, wts, 1.0, N) as eng:
        F, G = eng.eval(x)
    assert_parity(F, G, F_ref, G_ref, n, what="64 x 1 states")


def _rank_one_sparse_problem(n, K, N, E, sys_type, herm, pairs, seed, shared):
    """pure states (sandwich: Xi = v v', Xt = w w'; left multiplication: n x 1 columns), control operators with a few
    symmetric pairs of non-zeros -- the shape of a three-qubit Liouville-space transfer (64 x 64, Pauli-type controls)"""
    rng = np.random.default_rng(seed)

    def gen():
        M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        return 0.2 * ((M + M.conj().T) / 2 if herm else M)
    A = np.array([gen() for _ in range(E)])
    B = np.zeros((E, K, n, n), complex)
    for k in range(E):
        for c in range(K):
            for _ in range(pairs):
                i, j = rng.integers(0, n, 2)
                v = rng.standard_normal() + 1j * rng.standard_normal()
                if i == j:
                    B[k, c, i, i] = v.real
                elif herm:
                    B[k, c, i, j], B[k, c, j, i] = v, np.conj(v)
                else:
                    B[k, c, i, j] = v
    if shared:
        B[:] = B[0]

    def vec():
        v = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        return v / np.linalg.norm(v)
    if sys_type == "UnitaryGate":
        Xi = np.array([vec().reshape(n, 1) for _ in range(E)])
        Xt = np.array([vec().reshape(n, 1) for _ in range(E)])
    else:
        Xi = np.array([np.outer(v, v.conj()) for v in (vec() for _ in range(E))])
        Xt = np.array([np.outer(v, v.conj()) for v in (vec() for _ in range(E))])
    return A, B, Xi, Xt, rng.uniform(0.2, 1.0, E), rng.uniform(-1, 1, (K, N))


@pytest.mark.parametrize("n,sys_type,K,N,E,pairs,herm,shared", [
    (64, "CoherenceTransfer", 6, 40, 3, 30, False, True), (48, "StateTransfer", 11, 7, 2, 20, True, False),
    (33, "UnitaryGate", 3, 1, 2, 12, True, False), (64, "UnitaryGate", 16, 33, 3, 60, False, True),
    (57, "StateTransfer", 4, 2, 2, 100, True, True), (40, "UnitaryGate", 1, 101, 1, 5, False, False),
    (64, "StateTransfer", 2, 64, 5, 126, False, False)])
@pytest.mark.parametrize("variant", [0, 1])
def test_grid_family_rank_one_states_run_on_vectors(qoc, oracle, monkeypatch, n, sys_type, K, N, E, pairs, herm, shared, variant):
    """grid_thin_kernel: v_{t+1} = P_t v_t, w_t = P_t' w_{t+1}, the gradient from the controls' (coefficient, position)
    lists -- against the oracle's DENSE evaluation of the same inputs at the 1e-10 bar, member by member, and against the
    library's own dense chain (GRAPE_NO_THIN=1); dense control operators keep the dense chain."""
    A, B, Xi, Xt, wts, x = _rank_one_sparse_problem(n, K, N, E, sys_type, herm, pairs, seed=11 * n + K + N, shared=shared)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, 1.1, variant=variant, per_member=True)
    with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.1, N, variant=variant, member_results=True, max_batch=2) as eng:
        info = eng.info
        F, G = eng.eval(x)
        foms, grads = eng.member_results()
        names = eng.kernel_names()
        P = eng.trajectory(E - 1, states=False)[0]
        Fb, Gb = eng.eval_batch(np.array([0.5 * x, x]))
    assert info["rank_one_chain"] == 1 and info["sparse_controls"] == 1 and info["states_stored"] == 0, info
    assert any(k.startswith("grid_thin_kernel") for k in names) and not any(k.startswith("grid_chain_kernel") for k in names), names
    for k in range(E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"member {k}")
    assert_parity(F, G, F_ref, G_ref, n, what="ensemble")
    assert Fb[1] == F and np.array_equal(Gb[1], G)
    if sys_type == "UnitaryGate":
        P_ref = oracle.member_eval_rect(A[-1], B[-1], Xi[-1], Xt[-1], x, 1.1, variant=variant, trajectory=True)[2]
    else:
        P_ref = oracle.member_eval(sys_type, A[-1], B[-1], Xi[-1], Xt[-1], x, 1.1, variant=variant, trajectory=True)[2]
    assert np.abs(P - P_ref).max() <= 1e-12 * max(1.0, np.abs(P_ref).max())
    monkeypatch.setenv("GRAPE_NO_THIN", "1")
    with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.1, N, variant=variant) as eng:
        assert eng.info["rank_one_chain"] == 0
        F_d, G_d = eng.eval(x)
    assert_parity(F, G, F_d, G_d, n, what="vector chain vs dense chain")


def test_grid_family_rank_one_states_with_dense_controls_keep_the_dense_chain(qoc, oracle):
    w = _random_problem(qoc, 40, 2, 6, 2, "StateTransfer", seed=3, hermitian=True)       # pure states, dense B
    w.A *= 0.2
    w.B *= 0.2
    with _engine(qoc, w) as eng:
        assert eng.info["rank_one_chain"] == 0 and eng.info["states_stored"] == 1
        F, G = eng.eval(w.x)
    F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T)
    assert_parity(F, G, F_ref, G_ref, w.n, what="dense controls")


def test_grid_family_on_a_group(qoc, oracle):
    """three shards on one GPU (peer sum): the ensemble axis splits as for every other family"""
    w = _random_problem(qoc, 40, 2, 7, 5, "UnitaryGate", seed=77, hermitian=True)
    w.A *= 0.2
    w.B *= 0.2
    F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T)
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, devices=[0, 0, 0],
                         flags=qoc.engine.FLAG_GROUP_PEER_SUM) as eng:
        F, G = eng.eval(w.x)
    assert_parity(F, G, F_ref, G_ref, w.n, what="group of three shards")


@pytest.mark.parametrize("n,sys_type,herm", [(7, "StateTransfer", True), (16, "CoherenceTransfer", False), (29, "UnitaryGate", True)])
def test_small_sizes_through_the_grid_kernels(qoc, oracle, monkeypatch, n, sys_type, herm):
    """GRAPE_GRID=1: NT = 1, 2 instances of the same kernels, against the oracle and against the tile family's result"""
    w = _random_problem(qoc, n, 3, 11, 3, sys_type, seed=400 + n, hermitian=herm, mixed=True)
    with _engine(qoc, w) as eng:
        F_tile, G_tile = eng.eval(w.x)
    monkeypatch.setenv("GRAPE_GRID", "1")
    F, G = _check(qoc, oracle, w)
    assert abs(F - F_tile) <= 1e-12 * max(1.0, abs(F_tile)) and np.abs(G - G_tile).max() <= 1e-12 * max(1.0, np.abs(G_tile).max())


@pytest.mark.parametrize("n,herm,sys_type,scale", [(40, True, "UnitaryGate", 0.25), (48, False, "StateTransfer", 0.25),
                                                    (64, True, "CoherenceTransfer", 0.2), (33, True, "UnitaryGate", 4.0)])
@pytest.mark.parametrize("objective", ["fom", "c1"])
def test_grid_family_exact_gradient(qoc, oracle, n, herm, sys_type, scale, objective):
    """the exact gradient / ADGRAPE functional at n = 33..64 (grid_exact_kernel: exact_tile.hip's one-derivative-per-slice
    algorithm on workgroup-owned matrices) against the oracle's independent restatement (block-triangular Pade), incl. the
    squaring path (scale 4) and the mixed states whose W2 differs from W1."""
    w = _random_problem(qoc, n, 2, 4, 2, sys_type, seed=1300 + n, hermitian=herm, mixed=True)
    w.A *= scale
    w.B *= scale
    obj = 0 if objective == "fom" else 1
    F_ref, G_ref = oracle.ensemble_exact(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T, variant=1, objective=obj)
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, variant=1, gradient="exact", objective=objective) as eng:
        F, G = eng.eval(w.x)
        names = eng.kernel_names()
    assert "grid_exact_kernel" in names
    assert_parity(F, G, F_ref, G_ref, n, what=f"exact gradient n={n} {objective}")


@pytest.mark.parametrize("n,sys_type,herm,variant,N,E,sparse", [(40, "UnitaryGate", True, 0, 37, 1, False),
                                                               (64, "UnitaryGate", True, 1, 48, 2, True),
                                                               (48, "StateTransfer", True, 0, 30, 1, False),
                                                               (33, "CoherenceTransfer", False, 0, 26, 3, False),
                                                               (64, "StateTransfer", True, 1, 24, 1, True)])
def test_grid_family_chunked_time_axis(qoc, oracle, monkeypatch, n, sys_type, herm, variant, N, E, sparse):
    """Round 6 (VERDICT r5 #4b; the single-`Problem` closure src/solve.jl:63-143 at n = 33..64): fewer members than half the
    compute units cut the time axis into chunks -- chunk products (grid_chunk_product_kernel), the boundary scan (the chain
    kernel on the chunk products), one workgroup per (member, chunk).  Every member against the oracle, propagators / states /
    costates of the stored trajectory, the sequential chain (GRAPE_NO_TP=1) to rounding, bitwise reproducible; ragged last
    chunk (N not a multiple of the chunk length), forced chunk counts."""
    if sparse:
        w = _sparse_problem(qoc, n, 4, N, E, sys_type, seed=40 + n, nnz_pairs=14)       # (Hermitian generators)
    else:
        w = _random_problem(qoc, n, 3, N, E, sys_type, seed=300 + n + N, hermitian=herm, mixed=True)
    w.A *= 0.2
    w.B *= 0.2
    res = {}
    for tag, env in (("chunked", {}), ("chunks5", {"GRAPE_TP_CHUNKS": "5"}), ("sequential", {"GRAPE_NO_TP": "1"})):
        for kk in ("GRAPE_TP_CHUNKS", "GRAPE_NO_TP"):
            monkeypatch.delenv(kk, raising=False)
        for kk, v in env.items():
            monkeypatch.setenv(kk, v)
        F, G = _check(qoc, oracle, w, variant=variant)
        with _engine(qoc, w, variant=variant) as eng:
            F1, G1 = eng.eval(w.x)
            F2, G2 = eng.eval(w.x)
            names = eng.kernel_names()
            chunks = eng.info["time_chunks"]
        assert F1 == F2 and np.array_equal(G1, G2)
        assert ("grid_chunk_product_kernel" in names) == (tag != "sequential"), names
        assert (chunks >= 2) == (tag != "sequential") and (tag != "chunks5" or chunks == 5)
        res[tag] = G1
    scale = np.abs(res["sequential"]).max()
    assert np.abs(res["chunked"] - res["sequential"]).max() <= 1e-11 * scale
    assert np.abs(res["chunks5"] - res["sequential"]).max() <= 1e-11 * scale
