"""GPU: edge shapes of the hot path -- single slice, ragged slice counts around the 64-lane
chunking, one control / many controls, 3x3 operators, single member, forced decompositions,
weights that are not normalised -- all against the oracle at the 1e-10 parity bar."""
import numpy as np
import pytest

from conftest import assert_parity

pytestmark = pytest.mark.gpu


def _problem(qoc, n, K, N, E, sys_type, seed):
    rng = np.random.default_rng(seed)

    def herm():
        M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        return (M + M.conj().T) / 2
    A = np.array([herm() for _ in range(E)])
    B = np.array([[herm() for _ in range(K)] for _ in range(E)]) * 0.5
    if sys_type == "UnitaryGate":
        Xi = np.array([np.eye(n, dtype=complex)] * E)
        Xt = np.array([np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))[0] for _ in range(E)])
    else:
        def rho():
            v = rng.standard_normal(n) + 1j * rng.standard_normal(n)
            v /= np.linalg.norm(v)
            return np.outer(v, v.conj())
        Xi = np.array([rho() for _ in range(E)])
        Xt = np.array([rho() for _ in range(E)])
    return qoc.workloads.Workload("edge", sys_type, n, K, N, E, 1.3, A, B, Xi, Xt, rng.uniform(0.2, 1.7, E),
                                  rng.uniform(-1, 1, (K, N)))


SHAPES = [  # n, K, N, E, engine kwargs
    (2, 1, 1, 1, {}), (2, 2, 63, 2, {}), (2, 2, 64, 2, {}), (2, 2, 65, 2, {}), (2, 3, 1000, 1, {}),
    (2, 2, 1025, 1, {}), (3, 2, 50, 3, {}), (3, 5, 200, 2, {"waves_per_member": 3}), (4, 1, 7, 5, {}),
    (4, 7, 129, 3, {}), (4, 4, 500, 2, {"waves_per_member": 4}), (4, 2, 300, 2, {"slices_per_lane": 9}),
    (4, 3, 64, 70, {}),
    # long pulses: the LDS staging buffer of controls/gradient no longer fits 4 members (fewer
    # members per workgroup), then not even one (global-scratch fallback)
    (4, 4, 1500, 8, {"waves_per_member": 1}), (2, 2, 20000, 1, {}), (4, 3, 9000, 2, {"waves_per_member": 1}),
]


@pytest.mark.parametrize("n,K,N,E,ekw", SHAPES)
@pytest.mark.parametrize("sys_type", ["UnitaryGate", "StateTransfer"])
@pytest.mark.parametrize("flow", ["auto", "general"])
def test_edge_shapes(qoc, oracle, n, K, N, E, ekw, sys_type, flow):
    w = _problem(qoc, n, K, N, E, sys_type, seed=n * 1000 + K * 100 + N + E)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T,
                                                            per_member=True)
    flags = 0 if flow == "auto" else qoc.engine.FLAG_FORCE_GENERAL
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, flags=flags, member_results=True, **ekw) as eng:
        F, G = eng.eval(w.x)
        foms, grads = eng.member_results()
    for k in range(E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"member {k}")
    assert_parity(F, G, F_ref, G_ref, n, what="ensemble")


def test_repeated_evaluations_and_new_controls(qoc, oracle):
    """the context is reused across evaluations like the reference's closure across optimiser steps."""
    w = qoc.workloads.config("C3", E=8, N=120)
    rng = np.random.default_rng(3)
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N) as eng:
        for _ in range(4):
            x = rng.uniform(-2, 2, w.x.shape)
            F, G = eng.eval(x)
            F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, x, w.T)
            assert_parity(F, G, F_ref, G_ref, w.n)


@pytest.mark.parametrize("n,E", [(2, 1), (3, 1), (4, 1), (4, 3), (2, 9)])
def test_single_workgroup_publishes_without_reduce_launch(qoc, oracle, monkeypatch, n, E):
    """an ensemble that fits ONE workgroup (single problems, src/solve.jl:63-143) has its weighted row written to the
    destination by the sweep kernel itself; same bits as through the reduce kernel, host and device entry points."""
    import torch
    w = _problem(qoc, n, 3, 200, E, "StateTransfer", seed=400 + n + E)
    F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("GRAPE_DIRECT_PUBLISH", mode)
        with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N) as eng:
            F, G = eng.eval(w.x)
            F2, G2 = eng.eval(w.x)
            xd = torch.as_tensor(np.ascontiguousarray(w.x.T), device="cuda")
            fg = torch.zeros(w.K * w.N + 1, dtype=torch.float64, device="cuda")
            eng.eval_device(xd.data_ptr(), fg.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            h = fg.cpu().numpy()
        assert F2 == F and np.array_equal(G, G2)
        assert h[-1] == F and np.array_equal(h[:-1].reshape(w.N, w.K).T, G)
        assert_parity(F, G, F_ref, G_ref, n, what=f"direct={mode}")
        res[mode] = (F, G)
    assert res["1"][0] == res["0"][0] and np.array_equal(res["1"][1], res["0"][1])


@pytest.mark.parametrize("n,where", [(4, "A_lower"), (4, "B_upper"), (16, "A_lower"), (16, "Xt")])
def test_non_finite_operator_entries_propagate(qoc, n, where):
    """a NaN anywhere in the operators must reach F and G, as it would in the reference's arithmetic -- in particular it
    must not be dropped by the Hermitian / rank-one shortcuts, which look at one triangle or one column only."""
    w = _problem(qoc, n, 2, 20, 2, "StateTransfer", seed=3)
    A, B, Xt = w.A.copy(), w.B.copy(), w.Xt.copy()
    if where == "A_lower":
        A[1, n - 1, 0] = np.nan
    elif where == "B_upper":
        B[0, 1, 0, n - 1] = np.nan
    else:
        Xt[1, n - 1, 1] = np.nan
    with qoc.GrapeEngine(w.sys_type, A, B, w.Xi, Xt, w.wts, w.T, w.N) as eng:
        F, G = eng.eval(w.x)
    assert np.isnan(F) and np.isnan(G).any()
