"""GPU: the size-generic path (sweep_any.hip) for operator dimensions outside the specialised families -- n = 1 and n > 64.
`_fom_and_gradient_GRAPE!` (src/GRAPE.jl:25-96) takes matrices of any size; so does grape_create now.  Correct at the 1e-10 bar
(every member's (F_k, g_k), the ensemble sums, propagators / states / costates to 1e-12), the reference's general flow.
Round 6: from n = 17 on its products run on the FP64 matrix cores (any_mm_mfma: 128 x 128 blocks of C, operands streamed
through LDS in k-panels; sizes that are and are not multiples of 16 / 32 / 128 below) and the propagators of an ensemble
smaller than the device are formed by several workgroups per member (any_prop_kernel); GRAPE_ANY_MFMA=0 / GRAPE_ANY_BLOCKS=1
keep round 5's scalar products / single launch -- all three agree."""
import numpy as np
import pytest

from conftest import assert_parity
from test_gpu_tile import _engine, _random_problem

pytestmark = pytest.mark.gpu


def _check(qoc, oracle, w, variant=0, **kw):
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T,
                                                            variant=variant, per_member=True)
    with _engine(qoc, w, variant=variant, flags=qoc.engine.FLAG_KEEP_COSTATES, **kw) as eng:
        F, G = eng.eval(w.x)
        foms, grads = eng.member_results()
        info = eng.info
        names = eng.kernel_names()
        k = w.E - 1
        P, X, L = eng.trajectory(k, costates=True)
        F2, G2 = eng.eval(w.x)
    assert info["kernel_family"] == 2 and info["states_stored"] == 1 and "any_sweep_kernel" in names
    assert F == F2 and np.array_equal(G, G2)                 # bitwise reproducible
    _, _, Pr, Xr, Lr = oracle.member_eval(w.sys_type, w.A[k], w.B[k], w.Xi[k], w.Xt[k], w.x, w.T, variant=variant, trajectory=True)
    for got, want, what in ((P, Pr, "propagators"), (X, Xr, "states"), (L, Lr, "costates")):
        assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), what
    if np.abs(grads_ref).max() < 1e-13:
        # (n = 1 with a real control operator: g = 2 Re(i dt b |x|^2 |l|^2) is zero identically -- both sides return rounding
        # noise of 1e-17, which no relative bar can compare)
        assert np.abs(grads).max() < 1e-13 and np.abs(G).max() < 1e-13
        assert np.abs(foms - foms_ref).max() <= 1e-10 * max(1.0, np.abs(foms_ref).max()) and abs(F - F_ref) <= 1e-10 * max(1.0, abs(F_ref))
        return F, G
    for m in range(w.E):
        assert_parity(foms[m], grads[m], foms_ref[m], grads_ref[m], w.n, what=f"n={w.n} member {m}")
    assert_parity(F, G, F_ref, G_ref, w.n, what=f"n={w.n}")
    return F, G


@pytest.mark.parametrize("n", [1, 65, 80, 96, 100, 128, 130])
@pytest.mark.parametrize("sys_type,herm", [("UnitaryGate", True), ("StateTransfer", False), ("CoherenceTransfer", True)])
@pytest.mark.parametrize("variant", [0, 1])
def test_any_size_random(qoc, oracle, n, sys_type, herm, variant):
    w = _random_problem(qoc, n, 2, 5, 2, sys_type, seed=900 + n, hermitian=herm, mixed=True)
    s = 0.6 if n == 1 else 2.0 / n
    w.A *= s
    w.B *= s
    _check(qoc, oracle, w, variant=variant)


def test_any_size_squarings_batches_chunks_groups(qoc, oracle, monkeypatch):
    w = _random_problem(qoc, 70, 3, 4, 5, "UnitaryGate", seed=12, hermitian=True)
    w.A *= 0.2                                               # dt |H|_1 of a few units: several squarings per slice
    w.B *= 0.2
    F0, G0 = _check(qoc, oracle, w)
    rng = np.random.default_rng(4)
    xs = np.array([w.x, rng.uniform(-1, 1, w.x.shape)])
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, max_batch=2) as eng:
        Fb, Gb = eng.eval_batch(xs)
    assert Fb[0] == F0 and np.array_equal(Gb[0], G0)
    Fr, Gr = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, xs[1], w.T)
    assert_parity(Fb[1], Gb[1], Fr, Gr, w.n, what="batch entry 1")
    monkeypatch.setenv("GRAPE_MAX_WORKSPACE_BYTES", str(2 * 2 * w.N * 70 * 70 * 16 + 1000))       # two members' P_t and X_t
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N) as eng:
        Fc, Gc = eng.eval(w.x)
        assert eng.info["member_chunk"] == 2
    assert Fc == F0 and np.array_equal(Gc, G0)               # member-chunked = unchunked, bit for bit
    monkeypatch.delenv("GRAPE_MAX_WORKSPACE_BYTES")
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, devices=[0, 0],
                         flags=qoc.engine.FLAG_GROUP_PEER_SUM) as eng:
        Fg, Gg = eng.eval(w.x)
    assert_parity(Fg, Gg, F0, G0, w.n, what="two shards")


def test_scalar_problem_through_solve(qoc):
    """n = 1: a one-level 'system' only collects a phase; UnitaryGate with Xt = e^{i phi}: F = Re(z^2) is minimised at -1."""
    prob = qoc.Problem(B=[np.array([[1.0 + 0j]])], A=np.array([[0.3 + 0j]]), Xi=np.array([[1.0 + 0j]]), Xt=np.array([[1j]]),
                       T=1.0, n_controls=1, guess=np.full((1, 6), 0.1), sys_type=qoc.UnitaryGate())
    sol = qoc.solve(prob, qoc.GRAPE(n_slices=6))
    assert sol.result.minimum <= 1.0


def test_exact_gradient_is_refused_outside_2_to_32(qoc):
    w = _random_problem(qoc, 70, 1, 3, 1, "UnitaryGate", seed=1)
    with pytest.raises(Exception) as ei:
        qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, gradient="exact")
    assert "2 <= n <= 64" in str(ei.value)


def test_matrix_core_products_agree_with_the_scalar_ones(qoc, oracle, monkeypatch):
    """any_mm_mfma against any_mm (GRAPE_ANY_MFMA=0) and the propagator launch against the single launch (GRAPE_ANY_BLOCKS=1)
    on a size that fills neither the last tile nor the last 32-block (n = 77), with squarings; kernel names asserted."""
    w = _random_problem(qoc, 77, 3, 9, 3, "StateTransfer", seed=31, hermitian=False, mixed=True)
    w.A *= 0.15
    w.B *= 0.15
    res = {}
    for tag, env in (("mfma", {}), ("scalar", {"GRAPE_ANY_MFMA": "0"}), ("one_launch", {"GRAPE_ANY_BLOCKS": "1"})):
        for kk in ("GRAPE_ANY_MFMA", "GRAPE_ANY_BLOCKS"):
            monkeypatch.delenv(kk, raising=False)
        for kk, v in env.items():
            monkeypatch.setenv(kk, v)
        with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N) as eng:
            res[tag] = eng.eval(w.x)
            names = eng.kernel_names()
        assert ("any_prop_kernel" in names) == (tag != "one_launch"), names
    F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T)
    for tag, (F, G) in res.items():
        assert_parity(F, G, F_ref, G_ref, w.n, what=tag)
    assert res["mfma"][0] == res["one_launch"][0] and np.array_equal(res["mfma"][1], res["one_launch"][1])      # the same products
    assert np.abs(res["mfma"][1] - res["scalar"][1]).max() <= 1e-12 * np.abs(res["scalar"][1]).max()


@pytest.mark.parametrize("n,sys_type,herm,variant,N,E", [(70, "UnitaryGate", True, 0, 26, 1), (96, "StateTransfer", False, 1, 20, 2),
                                                        (130, "CoherenceTransfer", True, 0, 17, 1)])
def test_chunked_time_axis_beyond_64(qoc, oracle, monkeypatch, n, sys_type, herm, variant, N, E):
    """Round 6: fewer members than compute units cut the time axis into chunks here too (sweep_any.hip phases 3-5: chunk
    products, the chain on the chunk products as the boundary scan, a workgroup per (member, chunk)) -- a single 128 x 128
    problem walked its slices in ONE workgroup on ONE compute unit.  Oracle parity incl. the stored trajectory, agreement with
    the sequential chain (GRAPE_NO_TP=1), forced chunk counts with a ragged last chunk, bitwise reproducible."""
    w = _random_problem(qoc, n, 3, N, E, sys_type, seed=60 + n, hermitian=herm, mixed=True)
    w.A *= 2.0 / n
    w.B *= 2.0 / n
    res = {}
    for tag, env in (("chunked", {}), ("chunks3", {"GRAPE_TP_CHUNKS": "3"}), ("sequential", {"GRAPE_NO_TP": "1"})):
        for kk in ("GRAPE_TP_CHUNKS", "GRAPE_NO_TP"):
            monkeypatch.delenv(kk, raising=False)
        for kk, v in env.items():
            monkeypatch.setenv(kk, v)
        _check(qoc, oracle, w, variant=variant)
        with _engine(qoc, w, variant=variant) as eng:
            F1, G1 = eng.eval(w.x)
            names = eng.kernel_names()
            chunks = eng.info["time_chunks"]
        assert ("any_chunk_product_kernel" in names and "any_scan_kernel" in names) == (tag != "sequential"), names
        assert (chunks >= 2) == (tag != "sequential") and (tag != "chunks3" or chunks == 3)
        res[tag] = G1
    scale = np.abs(res["sequential"]).max()
    assert np.abs(res["chunked"] - res["sequential"]).max() <= 1e-11 * scale
    assert np.abs(res["chunks3"] - res["sequential"]).max() <= 1e-11 * scale


@pytest.mark.parametrize("n,sys_type,herm,variant,E,N", [(66, "UnitaryGate", True, 0, 3, 7), (128, "UnitaryGate", True, 1, 2, 18),
                                                        (81, "StateTransfer", False, 1, 1, 6), (100, "CoherenceTransfer", True, 0, 2, 5)])
def test_sparse_shared_controls_beyond_64(qoc, oracle, monkeypatch, n, sys_type, herm, variant, E, N):
    """Round 6: shared control operators with few non-zeros (local drives on seven qubits and more: n of n^2 entries) -- the H
    build and the gradient traces of sweep_any.hip walk lists (grape_host::build_any_sparse: by element for H, by control for
    the traces) instead of K dense n x n operators per slice.  Oracle parity per member incl. the trajectory, `sparse_controls`
    reported, agreement with the dense walk (GRAPE_NO_SPARSE=1) to rounding; an operator set beyond the density bound and
    per-member operators keep the dense walk."""
    rng = np.random.default_rng(500 + n)
    K = 5
    w = _random_problem(qoc, n, K, N, E, sys_type, seed=70 + n, hermitian=herm, mixed=True)
    w.A *= 2.0 / n
    B = np.zeros((K, n, n), dtype=complex)
    for c in range(K):                                       # c = 0: diagonal; others: a few off-diagonals (Hermitian where asked),
        if c == 0:                                           # two controls sharing positions, one with a single entry
            B[c][np.arange(n), np.arange(n)] = rng.standard_normal(n)
            continue
        cnt = 1 if c == 4 else n
        ii, jj = rng.integers(0, n, cnt), rng.integers(0, n, cnt)
        if c == 3:
            ii, jj = np.nonzero(B[2])[0][:cnt], np.nonzero(B[2])[1][:cnt]
        v = rng.standard_normal(len(ii)) + 1j * rng.standard_normal(len(ii))
        B[c][ii, jj] = v
        if herm:
            B[c] = (B[c] + B[c].conj().T) / 2
    w.B = np.broadcast_to(B * 0.3, (E,) + B.shape).copy()
    res = {}
    for tag, env in (("lists", {}), ("dense", {"GRAPE_NO_SPARSE": "1"})):
        monkeypatch.delenv("GRAPE_NO_SPARSE", raising=False)
        for kk, v in env.items():
            monkeypatch.setenv(kk, v)
        _check(qoc, oracle, w, variant=variant)
        with _engine(qoc, w, variant=variant) as eng:
            res[tag] = eng.eval(w.x)
            assert eng.info["sparse_controls"] == (1 if tag == "lists" else 0)
    monkeypatch.delenv("GRAPE_NO_SPARSE", raising=False)
    assert abs(res["lists"][0] - res["dense"][0]) <= 1e-12 * max(1.0, abs(res["dense"][0]))
    assert np.abs(res["lists"][1] - res["dense"][1]).max() <= 1e-12 * np.abs(res["dense"][1]).max()
    # beyond the density bound (all operators together > n^2 / 4 non-zeros), and per-member operators: the dense walk
    w2 = _random_problem(qoc, n, 2, N, max(E, 2), sys_type, seed=71 + n, hermitian=herm, mixed=True)
    w2.A *= 2.0 / n
    w2.B *= 2.0 / n
    w2.B[:] = w2.B[0]
    with _engine(qoc, w2, variant=variant) as eng:
        eng.eval(w2.x)
        assert eng.info["sparse_controls"] == 0
    w3 = _random_problem(qoc, n, K, N, max(E, 2), sys_type, seed=72 + n, hermitian=herm, mixed=True)
    w3.A *= 2.0 / n
    w3.B = np.broadcast_to(B * 0.3, (max(E, 2),) + B.shape).copy()
    w3.B[1] *= 1.25
    with _engine(qoc, w3, variant=variant) as eng:
        eng.eval(w3.x)
        assert eng.info["sparse_controls"] == 0
    _check(qoc, oracle, w3, variant=variant)
