"""GPU: the process-per-GPU exchange without RCCL (grape_ipc_export / grape_ipc_attach, collective="ipc") between REAL
processes -- 2 and 4 ranks sharing the one GPU of the test box, which RCCL refuses -- and the in-process group's
arrive-and-sum.  Every rank must return the full-ensemble (F, G): parity with the oracle, bitwise equality between the
ranks, and bitwise equality with an in-process group context over the same shards (same rows, same order of summation)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_parity

pytestmark = pytest.mark.gpu


def _run_ranks(tmp_path, R, cfg, E, N, env=None, timeout=600):
    out = str(tmp_path / "ipc")
    port = 29600 + (os.getpid() + 7 * R) % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={R}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "workers", "ipc_rank.py"), out, cfg, str(E), str(N)]
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **(env or {}))
    p = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    return [np.load(f"{out}.rank{r}.npz") for r in range(R)]


@pytest.mark.parametrize("R,cfg,E,N", [(2, "C3", 10, 60), (4, "C3", 13, 40), (2, "C4", 6, 30), (3, "C1", 1, 10)])
def test_ranks_sharing_the_gpu_exchange_through_mailboxes(qoc, oracle, tmp_path, R, cfg, E, N):
    w = qoc.workloads.config(cfg, E=E, N=N) if cfg != "C1" else qoc.workloads.config(cfg)
    res = _run_ranks(tmp_path, R, cfg, w.E, w.N)
    if w.E < R:                                    # a rank without members: everybody agrees on the torch fallback
        assert all(str(r["collective"]) == "torch" for r in res)
        return
    assert all(str(r["collective"]) == "ipc" and int(r["comm_size"]) == R for r in res), [str(r["error"]) for r in res]
    rng = np.random.default_rng(5)
    xs = [w.x] + [w.x + 0.1 * rng.standard_normal(w.x.shape) for _ in range(4)]
    for i, x in enumerate(xs):
        F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, x, w.T)
        assert_parity(float(res[0]["F"][i]), res[0]["G"][i], F_ref, G_ref, w.n, what=f"evaluation {i}")
    for r in res[1:]:                              # every rank: the same bits
        assert np.array_equal(r["F"], res[0]["F"]) and np.array_equal(r["G"], res[0]["G"])
        assert np.array_equal(r["fg_device"], res[0]["fg_device"]) and r["F6"] == res[0]["F6"]
    # device entry point = host entry point on the same x; the exchange kernel closes the evaluation
    assert res[0]["fg_device"][-1] == res[0]["F"][1] and np.array_equal(res[0]["fg_device"][:-1].reshape(w.N, w.K).T, res[0]["G"][1])
    assert res[0]["F6"] == res[0]["F"][2] and np.array_equal(res[0]["G6"], res[0]["G"][2])
    assert str(res[0]["names"]).endswith("ipc_allreduce_kernel")
    # the in-process group over the same contiguous shards: same rows, same order -> the same bits
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, devices=[0] * R,
                         flags=qoc.engine.FLAG_GROUP_PEER_SUM) as eng:
        for i, x in enumerate(xs[:3]):
            F, G = eng.eval(x)
            assert F == res[0]["F"][i] and np.array_equal(G, res[0]["G"][i]), i


def test_device_lbfgs_over_the_mailbox_exchange(qoc, tmp_path):
    """grape_lbfgs on attached contexts: every evaluation of the device loop ends in the exchange kernel; all ranks walk the
    same iterates -- those of the single-context run (the reference's n_ens = 5 StateTransfer testset)."""
    w = qoc.workloads.reference_ensemble("StateTransfer", 5, 25, 5.0)
    res = _run_ranks(tmp_path, 2, "REF", 5, 25, env={"IPC_TEST_LBFGS": "1"})
    assert all(str(r["collective"]) == "ipc" for r in res)
    assert np.array_equal(res[0]["lbfgs_x"], res[1]["lbfgs_x"]) and res[0]["lbfgs_min"] == res[1]["lbfgs_min"]
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N) as eng:
        xm, info = eng.lbfgs(w.x, iterations=15)
    assert info["evaluations"] == int(res[0]["lbfgs_evals"])
    assert abs(info["minimum"] - float(res[0]["lbfgs_min"])) <= 1e-10 and np.abs(xm - res[0]["lbfgs_x"]).max() <= 1e-7


@pytest.mark.parametrize("G", [2, 5, 8])
def test_group_arrive_and_sum_equals_the_stream_ordered_reduction(qoc, oracle, monkeypatch, G):
    """In-process groups (round 4): every shard ends in shard_arrive_kernel on its own stream and the last blocks to arrive
    sum the rows -- against the round-3 path (events + reduce_shards_kernel on the first device, GRAPE_GROUP_STREAM_SUM=1):
    the same bits, many evaluations in a row (the counters must come back to zero every time)."""
    w = qoc.workloads.config("C3", E=19, N=50)
    rng = np.random.default_rng(3)
    xs = [w.x + 0.05 * rng.standard_normal(w.x.shape) for _ in range(6)]
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("GRAPE_GROUP_STREAM_SUM", mode)
        with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, devices=[0] * G,
                             flags=qoc.engine.FLAG_GROUP_PEER_SUM) as eng:
            got[mode] = [eng.eval(x) for x in xs for _ in range(3)]
            names = eng.kernel_names()
            assert (names[-1] == "shard_arrive_kernel") == (mode == "0"), names
    for (F0, G0), (F1, G1) in zip(got["0"], got["1"]):
        assert F0 == F1 and np.array_equal(G0, G1)
    F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, xs[-1], w.T)
    assert_parity(got["0"][-1][0], got["0"][-1][1], F_ref, G_ref, w.n, what="arrive-and-sum")


def test_batched_calls_over_the_mailbox_exchange(qoc, oracle, tmp_path):
    """grape_eval_batch on attached contexts (round 3: single-device only): n_x control arrays per call, one exchange."""
    w = qoc.workloads.config("C3", E=9, N=30)
    res = _run_ranks(tmp_path, 2, "C3", 9, 30, env={"IPC_TEST_BATCH": "3"})
    assert all(str(r["collective"]) == "ipc" for r in res)
    assert np.array_equal(res[0]["Fb"], res[1]["Fb"]) and np.array_equal(res[0]["Gb"], res[1]["Gb"])
    for b in range(3):                             # entry b of a batched call = the single call on the same x, bit for bit
        assert res[0]["Fb"][b] == res[0]["F"][b] and np.array_equal(res[0]["Gb"][b], res[0]["G"][b])
    assert res[0]["Fb1"][0] == res[0]["F"][1] and np.array_equal(res[0]["Gb1"][0], res[0]["G"][1])


@pytest.mark.parametrize("mode", ["arrive", "stream", "rccl1"])
def test_batched_calls_on_multi_device_contexts(qoc, oracle, monkeypatch, mode):
    """max_batch > 1 on in-process groups (arrive-and-sum, the stream-ordered reduction) and on a 1-rank RCCL communicator:
    host and device entry points against the oracle, and the batched ladder search of grape_lbfgs on a group."""
    import torch
    w = qoc.workloads.config("C3", E=11, N=40)
    rng = np.random.default_rng(8)
    X = np.array([w.x + 0.1 * rng.standard_normal(w.x.shape) for _ in range(3)])
    monkeypatch.setenv("GRAPE_GROUP_STREAM_SUM", "1" if mode == "stream" else "0")
    kw = dict(force_collective=True) if mode == "rccl1" else dict(devices=[0, 0, 0], flags=qoc.engine.FLAG_GROUP_PEER_SUM)
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, max_batch=3, **kw) as eng:
        Fb, Gb = eng.eval_batch(X)
        F1, G1 = eng.eval(X[1])
        F2, G2 = eng.eval_batch(X[:2])
        xd = torch.as_tensor(np.ascontiguousarray(np.swapaxes(X, 1, 2)), device="cuda")
        fg = torch.zeros(3 * (w.K * w.N + 1), dtype=torch.float64, device="cuda")
        eng.eval_batch_device(3, xd.data_ptr(), fg.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        h = fg.cpu().numpy().reshape(3, -1)
    for b in range(3):
        F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, X[b], w.T)
        assert_parity(Fb[b], Gb[b], F_ref, G_ref, w.n, what=f"{mode}: batch entry {b}")
        assert h[b, -1] == Fb[b] and np.array_equal(h[b, :-1].reshape(w.N, w.K).T, Gb[b])
    assert F1 == Fb[1] and np.array_equal(G1, Gb[1]) and np.array_equal(F2, Fb[:2]) and np.array_equal(G2, Gb[:2])


def test_ladder_search_on_a_group_with_batched_probes(qoc):
    """grape_lbfgs line_search = "ladder" (B step lengths per batched launch) on a multi-device context with max_batch >= 2
    (round 3: refused): the same iterates as on one device."""
    w = qoc.workloads.reference_ensemble("StateTransfer", 5, 25, 5.0)
    args = (w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N)
    with qoc.GrapeEngine(*args, max_batch=4) as eng:
        x1, i1 = eng.lbfgs(w.x, iterations=10, line_search="ladder", probes=4)
    with qoc.GrapeEngine(*args, max_batch=4, devices=[0, 0], flags=qoc.engine.FLAG_GROUP_PEER_SUM) as eng:
        x2, i2 = eng.lbfgs(w.x, iterations=10, line_search="ladder", probes=4)
    assert i1["probes"] == i2["probes"] == 4 and i1["iterations"] == i2["iterations"] and i1["evaluations"] == i2["evaluations"]
    assert abs(i1["minimum"] - i2["minimum"]) <= 1e-10 and np.abs(x1 - x2).max() <= 1e-7
