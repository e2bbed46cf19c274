"""Ensemble sharding across GPUs: one process per GPU, one all-reduce per evaluation.

The ensemble members are independent given the shared controls x
(/root/reference/src/solve.jl:166-187); the only coupling is the weighted sum of F and G
(:171-186, :191).  Each rank owns a contiguous block of members, evaluates it with its own
libgrape_hip context into a device buffer fg = [G (K*N), F], and a single
all_reduce(SUM) of those K*N+1 doubles over RCCL/xGMI completes the closure.  16 KB at the
headline config: latency-bound, one collective per optimiser step, nothing else crosses GPUs.

torch is used for what it is here for: device buffers, the current HIP stream and
torch.distributed (backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests).  The evaluator
is injectable so the world_size=2 gloo tests can exercise the sharding/reduction logic on CPU
tensors with a stand-in local evaluator; the product always passes GrapeEngine.
"""
import numpy as np


def shard_bounds(E, world_size, rank):
    """Contiguous blocks of ceil(E / world) members; trailing ranks may be empty."""
    per = -(-E // world_size)
    lo = min(E, rank * per)
    return lo, min(E, lo + per)


class ShardedGrape:
    """Evaluate F, G of an ensemble split over the ranks of a torch.distributed group.

    make_local(lo, hi) -> object with eval_device(d_x_ptr, d_fg_ptr, stream) for members [lo, hi)
    (or None when the shard is empty).  `device` is a torch.device."""

    def __init__(self, E, K, N, make_local, device, group=None, force_collective=False):
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

    def eval_device(self, x_dev):
        """x_dev: float64 tensor holding x as (K,N) column-major, i.e. shape (N, K) contiguous.
        Returns the all-reduced fg tensor (valid on the current stream)."""
        torch = self.torch
        if self.local is not None:
            stream = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0
            self.local.eval_device(x_dev.data_ptr(), self.fg.data_ptr(), stream)
        else:
            self.fg.zero_()
        if self.world > 1 or self.force_collective:
            self.dist.all_reduce(self.fg, op=self.dist.ReduceOp.SUM, group=self.group)
        return self.fg

    def eval(self, x):
        """Host convenience: x (K,N) numpy -> (F, G (K,N))."""
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


def sharded_engine(workload, device, group=None, force_collective=False, **engine_kw):
    """The product wiring: every rank builds a GrapeEngine for its block of `workload`."""
    from .engine import GrapeEngine

    w = workload
    dev_index = device.index if device.index is not None else 0

    def make_local(lo, hi):
        return GrapeEngine(w.sys_type, w.A[lo:hi], w.B[lo:hi], w.Xi[lo:hi], w.Xt[lo:hi], w.wts[lo:hi],
                           w.T, w.N, device=dev_index, **engine_kw)

    return ShardedGrape(w.E, w.K, w.N, make_local, device, group, force_collective)
