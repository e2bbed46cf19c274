"""GPU: grape_create keeps the lane kernels (n <= 4) at 16 slices per lane or fewer by giving long pulses more waves per member
(round 6; tools/w_sweep.py, profiles/r06_w_sweep.txt) -- an ensemble that fills the device alone used to get one wave per
member whatever N was.  The decomposition the rule picks, and parity of spot members against the oracle where it applies."""
import numpy as np
import pytest

from conftest import assert_parity

pytestmark = pytest.mark.gpu


def _ensemble(n, E, N, K, seed):
    rng = np.random.default_rng(seed)

    def herm():
        M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        return (M + M.conj().T) / 2
    A = np.array([herm() for _ in range(E)]) * 0.7
    B = np.broadcast_to(np.array([herm() for _ in range(K)]) * 0.5, (E, K, n, n)).copy()
    q, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    Xi = np.broadcast_to(np.eye(n, dtype=complex), (E, n, n)).copy()
    Xt = np.broadcast_to(q, (E, n, n)).copy()
    return A, B, Xi, Xt, np.full(E, 1.0 / E), rng.uniform(-1, 1, (K, N))


@pytest.mark.parametrize("n,E,N,W_want,S_want", [(4, 2100, 1100, 4, 9), (4, 1024, 500, 2, 8), (4, 1024, 1100, 4, 9), (2, 2100, 1100, 2, 9),
                                                  (3, 2100, 2100, 4, 9), (2, 8192, 500, 1, 8)])
def test_lane_kernels_keep_16_slices_per_lane(qoc, oracle, n, E, N, W_want, S_want):
    A, B, Xi, Xt, wts, x = _ensemble(n, E, N, 2, seed=40 + n + N)
    with qoc.GrapeEngine("UnitaryGate", A, B, Xi, Xt, wts, 2.0, N, member_results=True) as eng:
        F, G = eng.eval(x)
        foms, grads = eng.member_results()
        info = eng.info
    assert info["kernel_family"] == 0 and info["slices_per_lane"] <= 16
    assert (info["waves_per_member"], info["slices_per_lane"]) == (W_want, S_want), (info["waves_per_member"], info["slices_per_lane"])
    for k in (0, E // 2, E - 1):
        f_ref, g_ref = oracle.member_eval("UnitaryGate", A[k], B[k], Xi[k], Xt[k], x, 2.0)
        assert_parity(foms[k], grads[k], f_ref, g_ref, n, what=f"n={n} E={E} N={N} member {k}")
    assert abs(F - float(np.dot(wts, foms))) <= 1e-12 * max(1.0, abs(F))


def test_explicit_waves_per_member_is_kept(qoc):
    """`grape_config.waves_per_member` overrides the rule (as it overrides the fill-the-device choice)."""
    A, B, Xi, Xt, wts, x = _ensemble(4, 2100, 1100, 2, seed=7)
    with qoc.GrapeEngine("UnitaryGate", A, B, Xi, Xt, wts, 2.0, 1100, waves_per_member=1) as eng:
        eng.eval(x)
        assert eng.info["waves_per_member"] == 1 and eng.info["slices_per_lane"] == 35
