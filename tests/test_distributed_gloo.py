"""The N>1 path on CPU: world_size=2 gloo processes run the product's ShardedGrape
(shard partition + ONE all-reduce of [G, F]) with the oracle standing in for each rank's local
evaluator, and the result must equal the unsharded oracle evaluation."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, assert_parity


class OracleLocal:
    """local evaluator with the GrapeEngine.eval_device signature, CPU tensors + the C oracle."""

    def __init__(self, w, lo, hi, K, N):
        self.w, self.lo, self.hi, self.K, self.N = w, lo, hi, K, N

    def eval_device(self, x_ptr, fg_ptr, stream):
        import ctypes
        from oracle import grape_oracle
        w = self.w
        x = np.ctypeslib.as_array(ctypes.cast(x_ptr, ctypes.POINTER(ctypes.c_double)), shape=(self.N, self.K)).T
        s = slice(self.lo, self.hi)
        F, G = grape_oracle.ensemble_eval(w.sys_type, w.A[s], w.B[s], w.Xi[s], w.Xt[s], w.wts[s], x, w.T)
        fg = np.ctypeslib.as_array(ctypes.cast(fg_ptr, ctypes.POINTER(ctypes.c_double)), shape=(self.K * self.N + 1,))
        fg[:-1] = G.T.reshape(-1)
        fg[-1] = F


def _worker(rank, world, port, E, out, backend="gloo", collective="torch"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group(backend, rank=rank, world_size=world)
    import quoptimalcontrol_jl_amd as qoc
    from quoptimalcontrol_jl_amd.distributed import ShardedGrape
    w = qoc.workloads.config("C3", E=E, N=30)
    sg = ShardedGrape(w.E, w.K, w.N, lambda lo, hi: OracleLocal(w, lo, hi, w.K, w.N), torch.device("cpu"),
                      collective=collective)
    if collective == "lib":          # the stand-in evaluator cannot join a library communicator: every rank must
        assert sg.collective == "torch" and ("comm_attach" in sg.attach_error or "no members" in sg.attach_error or
                                             "another rank" in sg.attach_error)
    F, G = sg.eval(w.x)
    if rank == 0:
        np.save(out, np.concatenate([G.reshape(-1), [F], [sg.lo, sg.hi]]))
    dist.destroy_process_group()


@pytest.mark.parametrize("E,world", [(6, 2), (5, 2), (1, 2)])
def test_sharded_allreduce_matches_unsharded(tmp_path, oracle, qoc, E, world):
    out = str(tmp_path / "r0.npy")
    port = 29600 + (os.getpid() + 7 * E) % 300
    mp.spawn(_worker, args=(world, port, E, out), nprocs=world, join=True)
    got = np.load(out)
    w = qoc.workloads.config("C3", E=E, N=30)
    F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T)
    assert_parity(got[-3], got[:-3].reshape(w.K, w.N), F_ref, G_ref, w.n, what=f"E={E} world={world}")
    assert (got[-2], got[-1]) == (0, -(-E // world))


def test_library_collective_falls_back_consistently(tmp_path, oracle, qoc):
    """collective='lib' on ranks that cannot attach (here: a CPU stand-in evaluator; on a GPU node: fewer members
    than ranks, RCCL missing): all ranks agree on the torch.distributed fallback and the result is unchanged.
    gloo is also bench.py's control plane (unique-id broadcast, barrier, max-time reduction)."""
    out = str(tmp_path / "r0.npy")
    port = 29600 + (os.getpid() + 91) % 300
    mp.spawn(_worker, args=(2, port, 6, out, "gloo", "lib"), nprocs=2, join=True)
    got = np.load(out)
    w = qoc.workloads.config("C3", E=6, N=30)
    F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T)
    assert_parity(got[-3], got[:-3].reshape(w.K, w.N), F_ref, G_ref, w.n, what="lib->torch fallback")


class AttachableLocal(OracleLocal):
    """a stand-in that COULD join a library communicator (has comm_attach): a rank without members must stop the
    others from calling it -- they would wait in ncclCommInitRank for a rank that never comes"""
    calls = 0

    def comm_unique_id(self):
        return bytes(128)

    def comm_attach(self, token, rank, world):
        AttachableLocal.calls += 1
        raise AssertionError("comm_attach called although a rank has no members")


def _empty_rank_worker(rank, world, port, E, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import quoptimalcontrol_jl_amd as qoc
    from quoptimalcontrol_jl_amd.distributed import ShardedGrape
    w = qoc.workloads.config("C3", E=E, N=30)
    sg = ShardedGrape(w.E, w.K, w.N, lambda lo, hi: AttachableLocal(w, lo, hi, w.K, w.N), torch.device("cpu"), collective="lib")
    assert sg.collective == "torch" and AttachableLocal.calls == 0
    assert ("no members" in sg.attach_error) == (sg.local is None)
    F, G = sg.eval(w.x)
    if rank == 0:
        np.save(out, np.concatenate([G.reshape(-1), [F]]))
    dist.destroy_process_group()


@pytest.mark.parametrize("E,world", [(1, 2), (5, 4)])
def test_rank_without_members_sends_everybody_to_the_fallback_before_any_attach(tmp_path, oracle, qoc, E, world):
    """shard_bounds leaves trailing ranks empty (E = 1 on 2 ranks, E = 5 on 4); feasibility is agreed by an
    all-reduce(MIN) BEFORE the unique-id broadcast, so the collectives stay matched and nobody blocks in the attach."""
    out = str(tmp_path / "r0.npy")
    mp.spawn(_empty_rank_worker, args=(world, 29600 + (os.getpid() + 211 + E) % 300, E, out), nprocs=world, join=True)
    got = np.load(out)
    w = qoc.workloads.config("C3", E=E, N=30)
    F_ref, G_ref = oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T)
    assert_parity(got[-1], got[:-1].reshape(w.K, w.N), F_ref, G_ref, w.n, what=f"E={E} world={world}")


def _bcast_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from quoptimalcontrol_jl_amd.distributed import _bcast_bytes
    token = bytes(range(128)) if rank == 0 else None
    got = _bcast_bytes(dist, token, 128, None)
    if rank == 1:
        open(out, "wb").write(got)
    dist.destroy_process_group()


def test_unique_id_broadcast_reaches_the_other_rank(tmp_path):
    out = str(tmp_path / "tok.bin")
    mp.spawn(_bcast_worker, args=(2, 29600 + (os.getpid() + 173) % 300, out), nprocs=2, join=True)
    assert open(out, "rb").read() == bytes(range(128))


class MailboxLocal(OracleLocal):
    """a stand-in with the ipc_export / ipc_attach pair: records what the wiring hands it.  `fail_on`: the rank whose attach
    raises (a peer's handle it cannot open)."""
    fail_on = -1

    def ipc_export(self, n_ranks):
        self.exported = n_ranks
        return bytes([self.lo % 251]) * 64

    def ipc_attach(self, handles, rank, n_ranks):
        self.handles = list(handles)
        if rank == MailboxLocal.fail_on:
            raise RuntimeError("cannot open a peer's mailbox")

    @property
    def info(self):
        return {"comm_size": self.exported}


def _ipc_worker(rank, world, port, E, out, fail_on):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import quoptimalcontrol_jl_amd as qoc
    from quoptimalcontrol_jl_amd.distributed import ShardedGrape
    MailboxLocal.fail_on = fail_on
    w = qoc.workloads.config("C3", E=E, N=30)
    res = {"rank": rank}
    try:
        sg = ShardedGrape(w.E, w.K, w.N, lambda lo, hi: MailboxLocal(w, lo, hi, w.K, w.N), torch.device("cpu"), collective="ipc")
        res["collective"] = sg.collective
        if sg.local is not None and hasattr(sg.local, "handles"):
            res["handles"] = [h[0] for h in sg.local.handles]          # every rank's 64 bytes, in rank order
    except RuntimeError as exc:
        res["raised"] = str(exc)
    np.save(f"{out}.{rank}.npy", np.array([repr(res)]))
    dist.destroy_process_group()


@pytest.mark.parametrize("E,world,fail_on", [(6, 2, -1), (9, 3, -1), (1, 2, -1), (6, 2, 1)])
def test_mailbox_handles_are_all_gathered_and_every_step_is_agreed(tmp_path, qoc, E, world, fail_on):
    """collective='ipc' on the CPU control plane: every rank's 64-byte handle reaches every rank in rank order; a rank without
    members sends everybody to the torch fallback before anything is exported; a rank whose attach fails takes the whole group down with
    the same error on every rank (the contexts that did attach expect their peers at every evaluation)."""
    out = str(tmp_path / "ipc")
    mp.spawn(_ipc_worker, args=(world, 29600 + (os.getpid() + 37 * E + world) % 300, E, out, fail_on), nprocs=world, join=True)
    res = [eval(str(np.load(f"{out}.{r}.npy")[0])) for r in range(world)]      # noqa: S307 -- our own repr of a dict
    per = -(-E // world)
    if E < world:
        assert all(r["collective"] == "torch" and "handles" not in r for r in res)
    elif fail_on >= 0:
        assert all("raised" in r for r in res)                      # nobody carries on alone
        assert "this rank" in res[fail_on]["raised"] and all("another rank" in r["raised"] for i, r in enumerate(res) if i != fail_on)
    else:
        assert all(r["collective"] == "ipc" for r in res)
        want = [(i * per) % 251 for i in range(world)]
        assert all(r["handles"] == want for r in res)
