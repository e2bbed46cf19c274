"""GPU: the multi-GPU machinery behind the C ABI, exercised on ONE device (the metered box has one):
a 1-device group context (ncclCommInitAll + grouped ncclAllReduce inside grape_eval), a 1-rank
communicator attached to a plain context (grape_comm_unique_id / grape_comm_attach), the device-pointer
entry point with the in-library all-reduce, shard planning, and the timed-event ring."""
import ctypes as C

import numpy as np
import pytest

from conftest import assert_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,kw", [("C3", {"E": 12, "N": 70}), ("C1", {}), ("C4", {"E": 3, "N": 40})])
def test_group_context_one_device_runs_the_rccl_allreduce(qoc, oracle, name, kw):
    w = qoc.workloads.config(name, **kw)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T,
                                                             per_member=True)
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, force_collective=True,
                         member_results=True) as eng:
        info = eng.info
        assert info["n_devices"] == 1 and info["comm_size"] == 1
        F, G = eng.eval(w.x)
        F2, G2 = eng.eval(w.x)
        foms, grads = eng.member_results()
        P = eng.trajectory(w.E - 1, states=False)[0]
    assert_parity(F, G, F_ref, G_ref, w.n, what="group eval")
    assert F == F2 and np.array_equal(G, G2)
    for k in range(w.E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], w.n, what=f"member {k}")
    assert P.shape == (w.N, w.n, w.n)


def test_group_context_device_pointers(qoc, oracle):
    import torch
    w = qoc.workloads.config("C3", E=9, N=64)
    F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T)
    dev = torch.device("cuda", 0)
    xd = torch.as_tensor(np.ascontiguousarray(w.x.T), device=dev)
    fg = torch.zeros(w.K * w.N + 1, dtype=torch.float64, device=dev)
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, force_collective=True, device=0) as eng:
        eng.eval_device(xd.data_ptr(), fg.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize(dev)
        h = fg.cpu().numpy()
        F1, G1 = eng.eval(w.x)                      # host path right after the device path (stream ordering)
    assert_parity(h[-1], h[:-1].reshape(w.N, w.K).T, F_ref, G_ref, w.n, what="group eval_device")
    assert_parity(F1, G1, F_ref, G_ref, w.n, what="group eval after eval_device")


def test_comm_attach_single_rank(qoc, oracle):
    """one process per GPU, world of one: unique id -> attach -> every evaluation all-reduces in-library."""
    w = qoc.workloads.config("C3", E=10, N=50)
    F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T)
    token = qoc.GrapeEngine.comm_unique_id()
    assert len(token) == 128
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N) as eng:
        eng.comm_attach(token, 0, 1)
        assert eng.info["comm_size"] == 1 and eng.info["comm_rank"] == 0
        F, G = eng.eval(w.x)
        with pytest.raises(qoc.GrapeError):
            eng.comm_attach(token, 0, 1)            # already attached
    assert_parity(F, G, F_ref, G_ref, w.n, what="comm_attach eval")


def test_sharded_engine_library_collective(qoc, oracle):
    """the product wiring bench.py uses: ShardedGrape with collective='lib' and a forced 1-rank communicator."""
    import torch
    from quoptimalcontrol_jl_amd.distributed import sharded_engine
    w = qoc.workloads.config("C3", E=16, N=100)
    sg = sharded_engine(w, torch.device("cuda", 0), force_collective=True, collective="lib")
    assert sg.collective == "lib" and sg.comm_size == 1
    F, G = sg.eval(w.x)
    F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T)
    assert_parity(F, G, F_ref, G_ref, w.n, what="lib collective")
    sg.close()


def test_two_devices_requested_on_a_one_gpu_box_fails_loudly(qoc):
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a single-GPU box")
    w = qoc.workloads.config("C3", E=8, N=20)
    with pytest.raises(qoc.GrapeError) as ei:
        qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, devices=[0, 1])
    assert ei.value.status == -3 and "device_ids[1]" in str(ei.value)
    with pytest.raises(qoc.GrapeError) as ei:
        qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, devices=[0, 0])
    assert ei.value.status == -1


def test_timed_event_ring_wraps(qoc):
    """GRAPE_FLAG_TIME_KERNELS: more timed launches than the ring holds; the count and the sum survive."""
    w = qoc.workloads.config("C1")
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, flags=qoc.engine.FLAG_TIME_KERNELS) as eng:
        for _ in range(700):
            eng.eval(w.x)
        ms, n = eng.kernel_time(reset=True)
        assert n == 700 and 0.0 < ms < 700 * 1.0
        eng.eval(w.x)
        ms2, n2 = eng.kernel_time()
        assert n2 == 1 and ms2 < ms


@pytest.mark.parametrize("name,kw,objective,devices", [("C3", {"E": 7, "N": 33}, "fom", [0, 0, 0]), ("C3", {"E": 5, "N": 20}, "c1", [0, 0]),
                                                       ("C4", {"E": 5, "N": 12}, "fom", [0, 0]), ("C1", {}, "c1", [0, 0])])
def test_exact_gradient_on_multi_shard_group(qoc, oracle, name, kw, objective, devices):
    """gradient = exact (and the C1 functional of the ADGRAPE path) on a context that spans several shards: every shard runs
    its debug-flow sweep + exact kernel on its block of members, the rows meet in the group's sum like any other evaluation."""
    w = qoc.workloads.config(name, **kw)
    F_ref, G_ref = oracle.ensemble_exact(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T, variant=1,
                                         objective=0 if objective == "fom" else 1)
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, variant=1, gradient="exact", objective=objective,
                         devices=devices, flags=qoc.engine.FLAG_GROUP_PEER_SUM) as eng:
        per = -(-w.E // len(devices))
        assert eng.info["n_devices"] == min(len(devices), -(-w.E // per))
        F, G = eng.eval(w.x)
        F2, G2 = eng.eval(w.x)
    assert_parity(F, G, F_ref, G_ref, w.n, what=f"{name} exact {objective} on {len(devices)} shards")
    assert F == F2 and np.array_equal(G, G2)


@pytest.mark.parametrize("name,kw,devices", [("C3", {"E": 5, "N": 70}, [0, 0]), ("C3", {"E": 7, "N": 33}, [0, 0, 0]),
                                             ("C1", {}, [0, 0]), ("C4", {"E": 5, "N": 24}, [0, 0]), ("C4", {"E": 5, "N": 70}, [0, 0, 0]),
                                             ("C3", {"E": 9, "N": 40}, [0, 0, 0, 0, 0, 0, 0, 0])])
def test_multi_shard_group_on_one_gpu_with_peer_sum(qoc, oracle, name, kw, devices):
    """SEVERAL shards behind one context on the one GPU there is (GRAPE_FLAG_GROUP_PEER_SUM lets device_ids repeat and
    sums the shards' [G, F] on the first device instead of calling RCCL): contiguous ceil(E / G) blocks incl. empty
    trailing ones, operator slicing, x fan-out, the sum, accessor routing -- everything of the in-library multi-GPU
    path except the collective itself."""
    import torch
    w = qoc.workloads.config(name, **kw)
    F_ref, G_ref, foms_ref, grads_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T,
                                                             per_member=True)
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, devices=devices,
                         flags=qoc.engine.FLAG_GROUP_PEER_SUM, member_results=True) as eng:
        info = eng.info
        per = -(-w.E // len(devices))
        assert info["n_devices"] == min(len(devices), -(-w.E // per)) and info["members_first_device"] == min(per, w.E)
        F, G = eng.eval(w.x)
        F2, G2 = eng.eval(w.x)
        foms, grads = eng.member_results()
        P_last = eng.trajectory(w.E - 1, states=False)[0]      # lives on the last non-empty shard
        xd = torch.as_tensor(np.ascontiguousarray(w.x.T), device="cuda")
        fg = torch.zeros(w.K * w.N + 1, dtype=torch.float64, device="cuda")
        eng.eval_device(xd.data_ptr(), fg.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        h = fg.cpu().numpy()
        F3, G3 = eng.eval(0.5 * w.x)                            # host path right after the device path
    assert_parity(F, G, F_ref, G_ref, w.n, what="multi-shard eval")
    assert F == F2 and np.array_equal(G, G2)
    assert h[-1] == F and np.array_equal(h[:-1].reshape(w.N, w.K).T, G)
    for k in range(w.E):
        assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], w.n, what=f"member {k}")
    if name == "C3":
        P_ref = oracle.member_eval(w.sys_type, w.A[-1], w.B[-1], w.Xi[-1], w.Xt[-1], w.x, w.T, trajectory=True)[2]
        assert np.abs(P_last - P_ref).max() <= 1e-12
    F3_ref, G3_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, 0.5 * w.x, w.T)
    assert_parity(F3, G3, F3_ref, G3_ref, w.n, what="second control array")


def test_duplicate_devices_need_the_peer_sum_flag(qoc):
    w = qoc.workloads.config("C3", E=4, N=10)
    with pytest.raises(qoc.GrapeError) as ei:
        qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, devices=[0, 0])
    assert "duplicate" in str(ei.value)


def test_two_ranks_sharing_the_gpu_fall_back_consistently(qoc):
    """`python bench.py --gpus 2` on the one-GPU box (GRAPE_BENCH_SHARE_GPU=1: plumbing only): spawn -> rendezvous ->
    grape_comm_attach refused by RCCL (both ranks on one GPU) -> torch's nccl group probed and refused -> every rank
    agrees on the host-staged gloo sum; the JSON line must report both ranks and the right figure of merit."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GRAPE_BENCH_SHARE_GPU="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                          "--blocks", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["config"]["collective"] == "torch" and d["value"] > 0 and d["scaling"] == "strong"
    # N > 1 runs carry the weak-scaling companion (every rank a full 1024-member shard) next to the strong headline
    weak = d["extra"]["weak_scaling"]
    assert weak["ensemble_total"] == 2048 and weak["value"] > 0 and weak["scaling"] == "weak", weak
    # ... and the same strong-scaling step through the library's mailbox exchange, which ranks sharing a GPU CAN run
    assert d["extra"]["ipc_exchange"].get("value", 0) > 0, d["extra"]["ipc_exchange"]
    w = qoc.workloads.config("C3")
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N) as eng:
        F, _ = eng.eval(w.x)
    assert abs(d["F"] - F) <= 1e-12


def test_bench_with_the_mailbox_exchange_on_two_ranks(qoc):
    """`python bench.py --gpus 2 --collective ipc` with both ranks on the one GPU: the library's own exchange carries the
    headline step (n_gpus = 2 ranks that joined, collective "ipc"), F equals the single-context evaluation."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GRAPE_BENCH_SHARE_GPU="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--collective", "ipc", "--steps", "5",
                          "--warmup", "2", "--blocks", "1", "--no-cpu-baseline", "--no-extra"], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["config"]["collective"] == "ipc" and d["value"] > 0
    assert d["roofline"]["kernel"].endswith("ipc_allreduce_kernel")
    w = qoc.workloads.config("C3")
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N) as eng:
        F, _ = eng.eval(w.x)
    assert abs(d["F"] - F) <= 1e-12
