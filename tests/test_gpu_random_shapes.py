"""GPU: seeded random sweep over shapes, decompositions, system types, variants and data flows of
the small-operator kernels against the oracle (1e-10 parity bar)."""
import numpy as np
import pytest

from conftest import assert_parity
from test_gpu_edges import _problem

pytestmark = pytest.mark.gpu


def _cases(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        n = int(rng.choice([2, 3, 4]))
        K = int(rng.integers(1, 7))
        N = int(rng.choice([1, 2, 5, 17, 63, 64, 65, 100, 129, 257, 300, 511]))
        E = int(rng.choice([1, 2, 3, 5, 8, 13, 33]))
        wmax = {2: 16, 3: 8, 4: 4}[n]
        ekw = {}
        if rng.random() < 0.5:
            ekw["waves_per_member"] = int(rng.integers(1, wmax + 1))
        if rng.random() < 0.3:
            ekw["slices_per_lane"] = int(rng.integers(1, 12))
        out.append((i, n, K, N, E, str(rng.choice(["UnitaryGate", "StateTransfer", "CoherenceTransfer"])),
                    int(rng.integers(0, 2)), str(rng.choice(["auto", "general", "debug"])), ekw))
    return out


@pytest.mark.parametrize("i,n,K,N,E,sys_type,variant,flow,ekw", _cases(60, 2026))
def test_random_shape(qoc, oracle, i, n, K, N, E, sys_type, variant, flow, ekw):
    w = _problem(qoc, n, K, N, E, "UnitaryGate" if sys_type == "UnitaryGate" else "StateTransfer", seed=9000 + i)
    w.sys_type = sys_type
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T,
                                                            variant=variant, per_member=True)
    flags = {"auto": 0, "general": qoc.engine.FLAG_FORCE_GENERAL, "debug": qoc.engine.FLAG_KEEP_COSTATES}[flow]
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, variant=variant, flags=flags,
                         member_results=True, **ekw) as eng:
        F, G = eng.eval(w.x)
        foms, grads = eng.member_results()
    for k in range(E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"case {i} member {k}")
    assert_parity(F, G, F_ref, G_ref, n, what=f"case {i}")


@pytest.mark.parametrize("i,n,K,N,E,sys_type,variant,flow,ekw", [c for c in _cases(90, 777) if c[1] != 3][:48])
def test_random_shape_other_kernel(qoc, oracle, monkeypatch, i, n, K, N, E, sys_type, variant, flow, ekw):
    """the same sweep on the NON-default small-n kernel of each size: the lane-pair kernel for n = 2
    (up to 16 waves per member), the lane-per-chunk kernel for n = 4."""
    monkeypatch.setenv("GRAPE_SMALL_KERNEL", "pair" if n == 2 else "lane")
    if n == 4 and ekw.get("waves_per_member", 1) > 4:
        ekw = dict(ekw, waves_per_member=4)
    w = _problem(qoc, n, K, N, E, "UnitaryGate" if sys_type == "UnitaryGate" else "StateTransfer", seed=5000 + i)
    w.sys_type = sys_type
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T,
                                                            variant=variant, per_member=True)
    flags = {"auto": 0, "general": qoc.engine.FLAG_FORCE_GENERAL, "debug": qoc.engine.FLAG_KEEP_COSTATES}[flow]
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, variant=variant, flags=flags,
                         member_results=True, **ekw) as eng:
        assert eng.info["lane_pair"] == (1 if n == 2 else 0)
        F, G = eng.eval(w.x)
        foms, grads = eng.member_results()
    for k in range(E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"case {i} member {k}")
    assert_parity(F, G, F_ref, G_ref, n, what=f"case {i}")


@pytest.mark.parametrize("E,N,K,W", [(1, 1, 1, 0), (3, 63, 2, 8), (1000, 37, 7, 0), (2, 2000, 20, 1), (130, 500, 4, 0), (5, 1024, 3, 8)])
@pytest.mark.parametrize("sys_type", ["UnitaryGate", "StateTransfer"])
def test_pair_kernel_stress_shapes(qoc, oracle, E, N, K, W, sys_type):
    """the headline kernel (n = 4) at awkward sizes: one slice, more waves than needed, ensembles that do not divide
    the workgroup, controls/gradient too long for LDS (global scratch), eight waves per member."""
    w = _problem(qoc, 4, K, N, E, sys_type, seed=31 * E + N)
    F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T, n_threads=8)
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, waves_per_member=W) as eng:
        assert eng.info["lane_pair"] == 1
        F, G = eng.eval(w.x)
    assert_parity(F, G, F_ref, G_ref, 4, what=f"E={E} N={N} K={K} W={W}")
