"""GPU: few 32 x 32 units on the chunked unitary flow -- four waves per dependent product (csrc/sweep_coop.hip) -- against
the oracle at the 1e-10 bar and against the one-wave kernels it replaces (GRAPE_NO_COOP=1)."""
import numpy as np
import pytest

from conftest import assert_parity

pytestmark = pytest.mark.gpu


def _pauli_sum(rng, nq, terms):
    P = [np.eye(2), np.array([[0, 1], [1, 0]]), np.array([[0, -1j], [1j, 0]]), np.array([[1, 0], [0, -1]])]
    out = 0
    for _ in range(terms):
        M = np.array([[1.0 + 0j]])
        for _ in range(nq):
            M = np.kron(M, P[int(rng.integers(0, 4))])
        out = out + rng.uniform(0.3, 1.0) * M
    return out


def _problem(n, K, N, E, sys_type, seed):
    rng = np.random.default_rng(seed)
    nq = 5

    def herm():
        M = rng.standard_normal((32, 32)) + 1j * rng.standard_normal((32, 32))
        return ((M + M.conj().T) / 2)[:n, :n]
    A = np.array([herm() for _ in range(E)]) * 0.1
    B = np.array([[(0.4 * _pauli_sum(rng, nq, 2))[:n, :n] for _ in range(K)] for _ in range(E)])      # sparse, Hermitian
    if sys_type == "UnitaryGate":
        Xi = np.array([np.eye(n, dtype=complex)] * E)
        Xt = np.array([np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))[0] for _ in range(E)])
    else:
        def rho():
            vs = [rng.standard_normal(n) + 1j * rng.standard_normal(n) for _ in range(3)]
            return sum(w * np.outer(v, v.conj()) / np.vdot(v, v).real for w, v in zip((0.5, 0.3, 0.2), vs))
        Xi = np.array([rho() for _ in range(E)])
        Xt = np.array([rho() for _ in range(E)])
    return A, B, Xi, Xt, rng.uniform(0.3, 1.0, E), rng.uniform(-1, 1, (K, N))


@pytest.mark.parametrize("n,K,N,E,sys_type", [(32, 3, 64, 1, "UnitaryGate"), (32, 6, 200, 1, "UnitaryGate"),
                                              (24, 2, 40, 2, "UnitaryGate"), (17, 1, 33, 3, "UnitaryGate"),
                                              (32, 3, 64, 1, "StateTransfer"), (29, 2, 90, 2, "CoherenceTransfer"),
                                              (32, 4, 130, 1, "CoherenceTransfer")])
@pytest.mark.parametrize("variant", [0, 1])
def test_four_wave_products_match_oracle_and_one_wave_kernels(qoc, oracle, monkeypatch, n, K, N, E, sys_type, variant):
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, sys_type, seed=n + N + K)
    F_ref, G_ref = oracle.ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, 1.5, variant=variant)
    res = {}
    for mode in ("coop", "one-wave"):
        if mode == "one-wave":
            monkeypatch.setenv("GRAPE_NO_COOP", "1")
        with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.5, N, variant=variant) as eng:
            info = eng.info
            assert info["kernel_family"] == 1 and info["unitary_flow"] == 1 and info["time_chunks"] >= 2
            assert info["sparse_controls"] == 1 and info["rank_one_chain"] == 0
            res[mode] = eng.eval(x)
        assert_parity(res[mode][0], res[mode][1], F_ref, G_ref, n, what=mode)
    if sys_type == "UnitaryGate":            # same arithmetic, same order: bitwise (the sandwich sums z over the waves in another order)
        assert res["coop"][0] == res["one-wave"][0] and np.array_equal(res["coop"][1], res["one-wave"][1])
