"""GPU: member-invariant control operators in the n = 5..32 family (csrc/prop_hoist.hip) -- the control sum of
src/timeevolution.jl:105-107 formed once per slice (ctrl_sum_kernel) and A'_k added per member, squarings from the norm
bound |A'_k| + |Gc_t| -- against the oracle at the 1e-10 bar, against the library's own per-member build
(GRAPE_HOIST=0), for every chain that consumes the propagators (rank-one vector chain with and without the fused
forward pass, dense general chains, unitary chains, time chunks), both formula variants, the squaring path, batches,
and bitwise reproducibility (the kernel hands data between LDS stores written in assembly and matrix-core results)."""
import numpy as np
import pytest

from conftest import assert_parity

pytestmark = pytest.mark.gpu


def _problem(n, K, N, E, sand, herm_gen, seed, rank_one=True, scale=0.6, cols=None):
    rng = np.random.default_rng(seed)

    def gen(h):
        M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        return (M + M.conj().T) / 2 if h else M
    A = np.array([gen(herm_gen) for _ in range(E)]) * scale
    B1 = np.array([gen(True) for _ in range(K)]) * 0.4
    B = np.broadcast_to(B1, (E, K, n, n)).copy()                   # the SAME control operators for every member

    def vec():
        v = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        return v / np.linalg.norm(v)

    def rho():
        if rank_one:
            v = vec()
            return np.outer(v, v.conj())
        return sum(p * np.outer(v, v.conj()) for p, v in zip((0.6, 0.3, 0.1), (vec(), vec(), vec())))
    if sand:
        Xi = np.array([rho() for _ in range(E)])
        Xt = np.array([rho() for _ in range(E)])
    elif cols == 1:
        Xi = np.array([vec().reshape(n, 1) for _ in range(E)])
        Xt = np.array([vec().reshape(n, 1) for _ in range(E)])
    else:
        def unitary():
            q, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
            return q
        Xi = np.array([np.eye(n, dtype=complex) for _ in range(E)])
        Xt = np.array([unitary() for _ in range(E)])
    return A, B, Xi, Xt, rng.uniform(0.2, 1.0, E), rng.uniform(-1, 1, (K, N))


CASES = [  # n, K, N, E, sys_type, Hermitian drift, rank-one states, state columns (UnitaryGate)
    (16, 4, 37, 9, "CoherenceTransfer", False, True, None),        # rank-one vector chain, non-Hermitian generator (C4's shape)
    (16, 3, 8, 12, "StateTransfer", True, True, None),
    (12, 2, 21, 8, "CoherenceTransfer", False, True, None),        # zero-padded tile
    (16, 4, 64, 10, "UnitaryGate", False, True, 1),                # n x 1 states, left multiplication
    (16, 4, 33, 8, "CoherenceTransfer", False, False, None),       # mixed states: dense general chain (two-wave split)
    (16, 2, 19, 9, "UnitaryGate", True, False, None),              # unitary flow
    (9, 5, 12, 8, "StateTransfer", True, False, None),
    (16, 1, 1, 8, "CoherenceTransfer", False, True, None),         # a single slice
    (32, 6, 20, 8, "UnitaryGate", True, False, None),              # two-tile family: four waves per propagator (C5's shape)
    (32, 3, 7, 9, "StateTransfer", False, False, None),            # general flow, non-Hermitian generator
    (24, 2, 11, 8, "CoherenceTransfer", True, False, None),        # zero-padded second tile
    (17, 4, 5, 8, "UnitaryGate", False, False, None),
    (32, 2, 1, 8, "UnitaryGate", True, False, None),
]


@pytest.mark.parametrize("n,K,N,E,sys_type,herm_gen,rank_one,cols", CASES)
@pytest.mark.parametrize("variant", [0, 1])
def test_hoisted_control_sum_matches_oracle_and_per_member_build(qoc, oracle, monkeypatch, n, K, N, E, sys_type, herm_gen,
                                                                 rank_one, cols, variant):
    sand = sys_type != "UnitaryGate"
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, sand, herm_gen, seed=31 * n + N + K, rank_one=rank_one, cols=cols)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, 1.1, variant=variant,
                                                             per_member=True)
    res = {}
    for hoist in ("1", "0"):
        monkeypatch.setenv("GRAPE_HOIST", hoist)
        for fuse in (("GRAPE_FORCE_FUSE", "GRAPE_NO_FUSE") if (rank_one and (sand or cols == 1)) else ("",)):
            monkeypatch.delenv("GRAPE_FORCE_FUSE", raising=False)
            monkeypatch.delenv("GRAPE_NO_FUSE", raising=False)
            if fuse:
                monkeypatch.setenv(fuse, "1")
            with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.1, N, variant=variant, member_results=True) as eng:
                assert eng.info["hoisted_controls"] == int(hoist) and eng.info["kernel_family"] == 1
                F, G = eng.eval(x)
                foms, grads = eng.member_results()
                F2, G2 = eng.eval(x)
            assert F == F2 and np.array_equal(G, G2), "run-to-run reproducibility"
            for k in range(E):
                assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"hoist {hoist} {fuse} member {k}")
            assert_parity(F, G, F_ref, G_ref, n, what=f"hoist {hoist} {fuse} ensemble")
            res[(hoist, fuse)] = (F, G)
    keys = list(res)
    for key in keys[1:]:
        assert abs(res[key][0] - res[keys[0]][0]) <= 1e-12 * max(1.0, abs(res[keys[0]][0]))
        assert np.abs(res[key][1] - res[keys[0]][1]).max() <= 1e-12 * np.abs(res[keys[0]][1]).max() + 1e-15


@pytest.mark.parametrize("scale,N", [(6.0, 9), (40.0, 5)])
@pytest.mark.parametrize("n", [16, 32])
def test_hoisted_squaring_path_and_propagators(qoc, oracle, monkeypatch, scale, N, n):
    """dt |H| of 2..20: two to six squarings chosen from the norm BOUND; propagators against the oracle's Pade."""
    monkeypatch.setenv("GRAPE_HOIST", "1")
    K, E = 3, 8
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, True, True, seed=5, scale=scale)
    F_ref, G_ref = oracle.ensemble_eval("StateTransfer", A, B, Xi, Xt, wts, x, 1.0)
    with qoc.GrapeEngine("StateTransfer", A, B, Xi, Xt, wts, 1.0, N) as eng:
        assert eng.info["hoisted_controls"] == 1
        F, G = eng.eval(x)
        P = eng.trajectory(E - 1, states=False)[0]
    assert_parity(F, G, F_ref, G_ref, n, what="squaring path")
    P_ref = oracle.member_eval("StateTransfer", A[-1], B[-1], Xi[-1], Xt[-1], x, 1.0, trajectory=True)[2]
    assert np.abs(P - P_ref).max() <= 1e-11 * max(1.0, np.abs(P_ref).max())


def test_hoisted_batch_and_device_entry_points(qoc, oracle, monkeypatch):
    torch = pytest.importorskip("torch")
    monkeypatch.setenv("GRAPE_HOIST", "1")
    n, K, N, E = 16, 4, 29, 8
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, True, False, seed=77)
    xs = np.stack([x, 0.5 * x, -x])
    with qoc.GrapeEngine("CoherenceTransfer", A, B, Xi, Xt, wts, 0.9, N, max_batch=3) as eng:
        assert eng.info["hoisted_controls"] == 1
        Fb, Gb = eng.eval_batch(xs)
        F1, G1 = eng.eval(x)
        d_x = torch.tensor(np.asfortranarray(x).T.copy().reshape(-1), device="cuda")
        d_fg = torch.empty(K * N + 1, dtype=torch.float64, device="cuda")
        eng.eval_device(d_x.data_ptr(), d_fg.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        fg = d_fg.cpu().numpy()
    for b in range(3):
        Fr, Gr = oracle.ensemble_eval("CoherenceTransfer", A, B, Xi, Xt, wts, xs[b], 0.9)
        assert_parity(Fb[b], Gb[b], Fr, Gr, n, what=f"batch entry {b}")
    assert F1 == Fb[0] and np.array_equal(G1, Gb[0])
    assert fg[-1] == F1 and np.array_equal(fg[:-1].reshape(N, K).T, G1)


def test_operators_reupload_switches_between_hoisted_and_per_member(qoc, oracle):
    """grape_set_operators decides per upload: invariant B -> hoisted, member-dependent B -> per-member build."""
    n, K, N, E = 16, 2, 17, 8
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, True, False, seed=3)
    B2 = B.copy()
    B2[3, 1] *= 1.25
    with qoc.GrapeEngine("CoherenceTransfer", A, B, Xi, Xt, wts, 0.7, N) as eng:
        for Bk, want in ((B, 1), (B2, 0), (B, 1)):                 # (B2: the same kernels with the per-member control sum)
            eng.set_operators(A, Bk, Xi, Xt, wts)
            assert eng.info["hoisted_controls"] == want
            F, G = eng.eval(x)
            Fr, Gr = oracle.ensemble_eval("CoherenceTransfer", A, Bk, Xi, Xt, wts, x, 0.7)
            assert_parity(F, G, Fr, Gr, n, what=f"upload (hoisted {want})")


@pytest.mark.parametrize("n,K,N,E,sys_type,herm_gen,rank_one", [(16, 4, 37, 9, "CoherenceTransfer", False, True),
                                                               (16, 3, 20, 2, "UnitaryGate", True, False),
                                                               (32, 6, 12, 3, "UnitaryGate", True, False),
                                                               (24, 2, 9, 2, "StateTransfer", False, False)])
def test_per_member_controls_run_the_new_kernels_too(qoc, oracle, monkeypatch, n, K, N, E, sys_type, herm_gen, rank_one):
    """Members with their OWN control operators, genuinely different ones (one control of one member rescaled on its own): the
    round-3 expm kernels form the control sum themselves (no pre-pass); against the oracle and against the round-2 kernel."""
    sand = sys_type != "UnitaryGate"
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, sand, herm_gen, seed=n + 7 * N, rank_one=rank_one)
    B = B * (1.0 + 0.07 * np.arange(E))[:, None, None, None]        # B_k = (1 + eps_k) B ...
    B[E - 1, 0] *= 1.3                                              # ... and one control that breaks the pattern
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, 0.9, per_member=True)
    res = {}
    for hoist in (None, "0"):
        if hoist is None:
            monkeypatch.delenv("GRAPE_HOIST", raising=False)
        else:
            monkeypatch.setenv("GRAPE_HOIST", hoist)
        with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 0.9, N, member_results=True) as eng:
            assert eng.info["hoisted_controls"] == 0
            F, G = eng.eval(x)
            foms, grads = eng.member_results()
            F2, G2 = eng.eval(x)
        assert F == F2 and np.array_equal(G, G2)
        for k in range(E):
            assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"member {k} (GRAPE_HOIST={hoist})")
        assert_parity(F, G, F_ref, G_ref, n, what=f"ensemble (GRAPE_HOIST={hoist})")
        res[hoist] = (F, G)
    assert np.abs(res[None][1] - res["0"][1]).max() <= 1e-12 * np.abs(res["0"][1]).max() + 1e-15


SCALED_CASES = [  # n, K, N, E, sys_type, Hermitian drift, rank-one states, state columns, environment, a kernel the flow must run
    (16, 4, 37, 9, "CoherenceTransfer", False, True, None, {}, "prop_hoist1_kernel"),              # expm + vector chain
    (16, 4, 40, 70, "CoherenceTransfer", False, True, None, {"GRAPE_ACTION": "1"}, "action_parts_kernel"),   # vector flow (C4's)
    (16, 4, 40, 70, "CoherenceTransfer", False, True, None, {"GRAPE_ACTION": "1", "GRAPE_ACT_WHOLE": "1"}, "action_parts_kernel"),
    (24, 3, 20, 40, "CoherenceTransfer", False, True, None, {"GRAPE_ACTION": "1"}, "action_thin2_kernel"),   # 32 x 32 images
    (16, 4, 33, 12, "CoherenceTransfer", False, False, None, {"GRAPE_NO_TP": "1"}, "chain_tile_split_kernel"),  # expm inside the two-wave chain
    (16, 2, 19, 9, "UnitaryGate", True, False, None, {"GRAPE_NO_TP": "1"}, "prop_hoist1_kernel"),      # unitary flow
    (32, 6, 12, 3, "UnitaryGate", True, False, None, {"GRAPE_NO_TP": "1"}, "prop_hoist2_kernel"),      # four waves per propagator
    (32, 3, 7, 9, "StateTransfer", False, False, None, {"GRAPE_NO_TP": "1"}, "prop_hoist2_kernel"),
    (48, 3, 6, 4, "UnitaryGate", True, False, None, {}, "grid_prop_kernel"),                           # n > 32: the grid family
]


@pytest.mark.parametrize("n,K,N,E,sys_type,herm_gen,rank_one,cols,env,kernel", SCALED_CASES)
def test_scaled_per_member_controls_keep_the_hoisted_flows(qoc, oracle, monkeypatch, n, K, N, E, sys_type, herm_gen, rank_one,
                                                           cols, env, kernel):
    """VERDICT r5 #2 -- EnsembleProblem.B_g (src/problems.jl:33-41, applied in src/tools.jl:42-53) exists so that members can
    differ in their control operators; the case the reference itself writes down is amplitude inhomogeneity, B_g(k) =
    (1 + eps_k) B.  grape_set_operators detects B_{k,c} = s_k B_{0,c} and the flows built on the per-slice control sum keep
    it: ctrl_sum_kernel / action_rows_kernel run on member 0's operators, a member forms A'_k + s_k Gc_t.  Every member
    against the oracle; the kernels asserted; GRAPE_CTRL_SCALE=0 (the per-member build of round 3) agrees to rounding."""
    sand = sys_type != "UnitaryGate"
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, sand, herm_gen, seed=n + 3 * N + E, rank_one=rank_one, cols=cols)
    A *= 0.5
    B = B * (1.0 + 0.05 * (np.arange(E) / E - 0.5))[:, None, None, None]      # bench.py's C4pm / C5pm scaling
    for kk, v in env.items():
        monkeypatch.setenv(kk, v)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, 0.9, per_member=True)
    res = {}
    for mode in ("scaled", "own"):
        if mode == "own":
            monkeypatch.setenv("GRAPE_CTRL_SCALE", "0")
        else:
            monkeypatch.delenv("GRAPE_CTRL_SCALE", raising=False)
        try:
            eng = qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 0.9, N, member_results=True)
        except qoc.GrapeError:
            assert mode == "own"                                    # (a forced vector flow the per-member build does not serve)
            continue
        with eng:
            F, G = eng.eval(x)
            foms, grads = eng.member_results()
            F2, G2 = eng.eval(x)
            names = eng.kernel_names()
            hoisted = eng.info["hoisted_controls"]
        assert F == F2 and np.array_equal(G, G2)
        if mode == "scaled":
            assert kernel in names, names
            assert "ctrl_sum_kernel" in names or "action_rows_kernel" in names, names
            assert hoisted == 1 or "action_rows_kernel" in names
        else:
            assert "ctrl_sum_kernel" not in names and "action_rows_kernel" not in names, names
        for k in range(E):
            assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"member {k} ({mode})")
        assert_parity(F, G, F_ref, G_ref, n, what=f"ensemble ({mode})")
        res[mode] = G
    if "own" in res:
        assert np.abs(res["scaled"] - res["own"]).max() <= 1e-11 * np.abs(res["own"]).max() + 1e-15


def test_scaled_controls_detection_is_exact(qoc, oracle):
    """B_k = s_k B_0 is decided entry by entry after scaling: a single entry off by 1e-9 keeps the member's own operators; a
    member-chunked evaluation (workspace budget) still builds the control sum from the WHOLE ensemble's member 0."""
    n, K, N, E = 16, 3, 14, 10
    A, B, Xi, Xt, wts, x = _problem(n, K, N, E, True, False, seed=77)
    Bs = B * (1.0 + 0.1 * np.arange(E))[:, None, None, None]
    Bo = Bs.copy()
    Bo[4, 1, 2, 3] *= 1.0 + 1e-9
    for Bk, want in ((Bs, True), (Bo, False)):
        with qoc.GrapeEngine("CoherenceTransfer", A, Bk, Xi, Xt, wts, 0.7, N) as eng:
            F, G = eng.eval(x)
            names = eng.kernel_names()
        assert (("ctrl_sum_kernel" in names) or ("action_rows_kernel" in names)) == want, names
        Fr, Gr = oracle.ensemble_eval("CoherenceTransfer", A, Bk, Xi, Xt, wts, x, 0.7)
        assert_parity(F, G, Fr, Gr, n, what=f"scaled pattern {want}")


@pytest.mark.parametrize("n,sys_type,herm_gen,N,scale", [(32, "UnitaryGate", True, 9, 0.6), (21, "StateTransfer", False, 5, 0.6),
                                                        (32, "StateTransfer", True, 3, 8.0)])
def test_device_filling_32x32_ensembles_take_the_grid_expm_kernel(qoc, oracle, monkeypatch, n, sys_type, herm_gen, N, scale):
    """Round 5: at 32 x 32 (two tiles per side) with member-invariant controls and at least one unit per compute unit -- C5's
    shape -- the expm runs in sweep_grid.hip's grid_prop_kernel (a workgroup of four waves per propagator, operands through
    k-contiguous LDS images: 85.5 ms against prop_hoist2_kernel's 96-97 at C5) and hands the same P_t dumps to the tile
    family's chains; GRAPE_HOIST2=1 keeps prop_hoist2_kernel.  Against the oracle, and the two kernels against each other,
    incl. the squaring path (scale 8)."""
    import torch
    E = torch.cuda.get_device_properties(0).multi_processor_count + 3
    A, B, Xi, Xt, wts, x = _problem(n, 3, N, E, sys_type != "UnitaryGate", herm_gen, seed=7 * n + N, rank_one=False, scale=scale)
    res = {}
    for keep in (None, "1"):
        if keep is None:
            monkeypatch.delenv("GRAPE_HOIST2", raising=False)
        else:
            monkeypatch.setenv("GRAPE_HOIST2", keep)
        with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, 1.0, N, member_results=True) as eng:
            F, G = eng.eval(x)
            foms, grads = eng.member_results()
            names = eng.kernel_names()
            assert eng.info["hoisted_controls"] == 1
        res[keep] = (F, G, names)
    assert "grid_prop_kernel" in res[None][2] and "ctrl_sum_kernel" in res[None][2]
    assert "prop_hoist2_kernel" in res["1"][2] and "grid_prop_kernel" not in res["1"][2]
    assert abs(res[None][0] - res["1"][0]) <= 1e-12 * max(1.0, abs(res["1"][0]))
    assert np.abs(res[None][1] - res["1"][1]).max() <= 1e-12 * max(1.0, np.abs(res["1"][1]).max())
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, 1.0, per_member=True)
    for k in (0, 1, E // 2, E - 1):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=f"member {k}")
    assert_parity(res[None][0], res[None][1], F_ref, G_ref, n, what="ensemble, grid expm kernel")
