"""Ensemble sharding across GPUs: one process per GPU, one all-reduce per evaluation.

The ensemble members are independent given the shared controls x
(/root/reference/src/solve.jl:166-187); the only coupling is the weighted sum of F and G
(:171-186, :191).  Each rank owns a contiguous block of members, evaluates it with its own
libgrape_hip context, and a single all-reduce(SUM) of the K*N+1 doubles [G, F] over RCCL/xGMI
completes the closure.  16 KB at the headline config: latency-bound, one collective per optimiser
step, nothing else crosses GPUs.

Where the collective runs (`collective=`):
  "lib"    (product default) inside libgrape_hip.so: rank 0 creates an RCCL unique id
           (grape_comm_unique_id), torch.distributed only carries those 128 bytes to the other ranks,
           every rank calls grape_comm_attach, and from then on grape_eval / grape_eval_device end
           in ncclAllReduce on the evaluation's own stream.  grape_eval(x) is then the complete
           host -> N GPUs -> host closure on every rank.
  "ipc"    inside libgrape_hip.so without librccl: every rank exports a mailbox in its device memory
           (grape_ipc_export, a HIP IPC handle), torch.distributed all-gathers the 64-byte handles, every rank
           opens its peers' mailboxes (grape_ipc_attach), and from then on an evaluation ends in ONE kernel that
           stores the rank's row into every mailbox, waits (bounded) for the others' and sums in rank order.
           Works with ranks sharing a GPU -- which RCCL refuses -- so the process-per-GPU exchange is testable
           on a one-GPU box; results are bitwise those of an in-process group with the same shards.
  "torch"  torch.distributed.all_reduce on the fg tensor (backend "nccl" == RCCL on ROCm; "gloo" in
           the CPU tests, where a stand-in evaluator replaces the GPU).  Fallback when a rank cannot
           join the library communicator (e.g. fewer members than ranks).

torch is used for what it is here for: device buffers, the current HIP stream and
torch.distributed as the control plane.
"""
import numpy as np


def shard_bounds(E, world_size, rank):
    """Contiguous blocks of ceil(E / world) members; trailing ranks may be empty."""
    per = -(-E // world_size)
    lo = min(E, rank * per)
    return lo, min(E, lo + per)


def _bcast_bytes(dist, payload, nbytes, group):
    """rank 0's `payload` (bytes) to every rank, as a CPU uint8 tensor (gloo) or a device tensor (nccl)."""
    import torch

    backend = dist.get_backend(group)
    buf = torch.zeros(nbytes, dtype=torch.uint8)
    if payload is not None:
        buf[:] = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
    if "gloo" in backend:
        dist.broadcast(buf, src=0, group=group)
        return bytes(buf.numpy().tobytes())
    dev = torch.device("cuda", torch.cuda.current_device())
    dbuf = buf.to(dev)
    dist.broadcast(dbuf, src=0, group=group)
    return bytes(dbuf.cpu().numpy().tobytes())


class ShardedGrape:
    """Evaluate F, G of an ensemble split over the ranks of a torch.distributed group.

    make_local(lo, hi) -> object with eval_device(d_x_ptr, d_fg_ptr, stream) for members [lo, hi)
    (or None when the shard is empty).  `device` is a torch.device."""

    def __init__(self, E, K, N, make_local, device, group=None, force_collective=False, collective="torch"):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.group = group
        self.force_collective = force_collective      # run the all-reduce even with one rank (testing)
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.lo, self.hi = shard_bounds(E, self.world, self.rank)
        self.E, self.K, self.N = E, K, N
        self.device = device
        self.local = make_local(self.lo, self.hi) if self.hi > self.lo else None
        self.fg = torch.zeros(K * N + 1, dtype=torch.float64, device=device)
        self.collective = "torch"
        self.stage_through_host = False               # torch fallback on a gloo group with device tensors
        self.comm_size = self.world
        self.attach_timeout_s = 180.0
        if collective == "lib" and (self.world > 1 or force_collective):
            self._attach_library_communicator()
        elif collective == "ipc" and self.world > 1:
            self._attach_ipc_mailboxes()

    def _attach_ipc_mailboxes(self):
        """Mailbox exchange (grape_ipc_export / grape_ipc_attach).  Every step is agreed by all ranks (MIN of a 0/1 flag):
        one rank that cannot take part sends everybody to the torch.distributed fallback, nobody waits for a rank that left."""
        torch, dist = self.torch, self.dist

        def agree(flag_value):
            flag = torch.tensor([flag_value], dtype=torch.int32)
            if "gloo" not in dist.get_backend(self.group):
                flag = flag.to(self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            return int(flag.item())

        can = int(self.local is not None and hasattr(self.local, "ipc_export") and self.world <= 8)
        if not agree(can):
            self.attach_error = "a rank owns no members, has no ipc_export, or there are more than 8 ranks"
            return self._probe_torch_group()
        ok, handle = 1, b"\0" * 64
        try:
            handle = self.local.ipc_export(self.world)
        except Exception as exc:                      # noqa: BLE001 -- any failure means "fall back", consistently
            self.attach_error = repr(exc)
            ok = 0
        if not agree(ok):
            return self._probe_torch_group()
        buf = torch.frombuffer(bytearray(handle), dtype=torch.uint8).clone()
        on_dev = "gloo" not in dist.get_backend(self.group)
        if on_dev:
            buf = buf.to(self.device)
        gathered = [torch.zeros_like(buf) for _ in range(self.world)]
        dist.all_gather(gathered, buf, group=self.group)
        handles = [bytes(g.cpu().numpy().tobytes()) for g in gathered]
        try:
            self.local.ipc_attach(handles, self.rank, self.world)
        except Exception as exc:                      # noqa: BLE001
            self.attach_error = repr(exc)
            ok = 0
        if agree(ok):
            self.collective = "ipc"
            self.comm_size = self.local.info["comm_size"]
        else:
            # some ranks opened their peers' mailboxes and some could not: the contexts that did attach now expect every
            # peer at every evaluation -- no rank may carry on alone, so EVERY rank fails the same way
            raise RuntimeError("grape_ipc_attach failed on " + ("this rank: " + self.attach_error if not ok else "another rank") +
                               "; rebuild the engines with collective='torch' (or 'lib')")

    def _attach_library_communicator(self):
        """Every rank joins an RCCL communicator owned by libgrape_hip.so; all ranks agree on the
        outcome (one failed rank sends everybody to the torch.distributed fallback)."""
        torch, dist = self.torch, self.dist

        def agree(flag_value):
            """MIN over the ranks of a 0/1 flag (every rank calls it at the same point)."""
            if not (self.distributed and self.world > 1):
                return flag_value
            flag = torch.tensor([flag_value], dtype=torch.int32)
            if "gloo" not in dist.get_backend(self.group):
                flag = flag.to(self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            return int(flag.item())

        # 1. feasibility, agreed BEFORE any broadcast or attach: a rank without members (E = 5 on 4 ranks, E < world)
        #    or without a library evaluator cannot join, and the others must not wait in ncclCommInitRank for it
        can = int(self.local is not None and hasattr(self.local, "comm_attach"))
        if not can:
            self.attach_error = "this rank owns no members (E < world size) or its evaluator has no comm_attach"
        if not agree(can):
            ok = 0
            if can:
                self.attach_error = "another rank cannot join the library communicator"
        else:
            # 2. every rank can: rank 0's unique id to everybody, then ncclCommInitRank on every rank
            ok, stuck = 1, False
            try:
                token = self.local.comm_unique_id() if self.rank == 0 else None
                if self.distributed and self.world > 1:
                    token = _bcast_bytes(dist, token, 128, self.group)
                # ncclCommInitRank blocks until every rank has joined; should the bootstrap wedge (no route between
                # ranks, a rank that died), do not hang the job: give up after attach_timeout_s
                import threading
                box = {}

                def _attach():
                    try:
                        self.local.comm_attach(token, self.rank, self.world)
                        box["ok"] = True
                    except Exception as exc:          # noqa: BLE001
                        box["err"] = exc
                th = threading.Thread(target=_attach, daemon=True)
                th.start()
                th.join(self.attach_timeout_s)
                if th.is_alive():
                    stuck = True
                    raise TimeoutError(f"grape_comm_attach did not return within {self.attach_timeout_s} s")
                if "err" in box:
                    raise box["err"]
            except Exception as exc:                  # noqa: BLE001 -- any failure means "fall back", consistently
                self.attach_error = repr(exc)
                ok = 0
            ok = agree(ok)
            any_stuck = not agree(0 if stuck else 1)  # agreed as well: the job fails collectively, nobody half-continues
            if any_stuck:
                # the abandoned thread is still inside ncclCommInitRank ON THAT RANK'S CONTEXT and may complete later: a
                # context that can grow a communicator behind our back must not evaluate (one-sided all-reduce), and the
                # other ranks must not walk into a fallback collective that the stuck rank will never join
                raise RuntimeError("grape_comm_attach timed out on " + ("this rank" if stuck else "another rank") +
                                   "; every rank aborts instead of evaluating: " + getattr(self, "attach_error", ""))
        if ok:
            self.collective = "lib"
            self.comm_size = self.local.info["comm_size"]
        else:
            self._probe_torch_group()

    def _probe_torch_group(self):
        """The torch.distributed fallback on a gloo control plane: give the data path its own RCCL group (torch's) and PROBE it --
        RCCL refuses, at the first collective, ranks that share a GPU; all ranks then agree to stage through the host."""
        torch, dist = self.torch, self.dist
        if self.distributed and self.world > 1 and self.device.type == "cuda" and "nccl" not in dist.get_backend(self.group):
            control = self.group
            works = 1
            nccl_group = None
            try:
                nccl_group = dist.new_group(backend="nccl")
                dist.all_reduce(torch.zeros(1, device=self.device), group=nccl_group)
                torch.cuda.synchronize(self.device)
            except Exception as exc:                  # noqa: BLE001 -- keep the (slow but correct) gloo path
                works = 0
                self.attach_error = getattr(self, "attach_error", "") + f"; nccl group: {exc!r}"
            flag = torch.tensor([works], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=control)
            if int(flag.item()):
                self.group = nccl_group
            else:
                self.stage_through_host = True

    def eval_device(self, x_dev):
        """x_dev: float64 tensor holding x as (K,N) column-major, i.e. shape (N, K) contiguous.
        Returns the all-reduced fg tensor (valid on the current stream)."""
        torch = self.torch
        if self.local is not None:
            stream = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0
            self.local.eval_device(x_dev.data_ptr(), self.fg.data_ptr(), stream)
        else:
            self.fg.zero_()
        if self.collective == "torch" and (self.world > 1 or self.force_collective):
            if self.stage_through_host:
                if self.device.type == "cuda":
                    torch.cuda.current_stream(self.device).synchronize()
                h = self.fg.cpu()
                self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
                self.fg.copy_(h)
            else:
                self.dist.all_reduce(self.fg, op=self.dist.ReduceOp.SUM, group=self.group)
        return self.fg

    def eval(self, x):
        """Host -> GPUs -> host: x (K,N) numpy -> (F, G (K,N)), the full-ensemble closure on every rank."""
        if self.collective in ("lib", "ipc") or (self.world == 1 and not self.force_collective and self.local is not None
                                                 and hasattr(self.local, "eval")):
            return self.local.eval(x)                 # grape_eval: the all-reduce happens inside the library
        torch = self.torch
        xd = torch.as_tensor(np.ascontiguousarray(np.asarray(x, float).T), device=self.device)
        fg = self.eval_device(xd)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        h = fg.cpu().numpy()
        return float(h[-1]), np.ascontiguousarray(h[:-1].reshape(self.N, self.K).T)

    def close(self):
        if self.local is not None and hasattr(self.local, "close"):
            self.local.close()
            self.local = None


def sharded_engine(workload, device, group=None, force_collective=False, collective="lib", **engine_kw):
    """The product wiring: every rank builds a GrapeEngine for its block of `workload`."""
    from .engine import GrapeEngine

    w = workload
    dev_index = device.index if device.index is not None else 0

    def make_local(lo, hi):
        return GrapeEngine(w.sys_type, w.A[lo:hi], w.B[lo:hi], w.Xi[lo:hi], w.Xt[lo:hi], w.wts[lo:hi],
                           w.T, w.N, device=dev_index, **engine_kw)

    return ShardedGrape(w.E, w.K, w.N, make_local, device, group, force_collective, collective)
