"""One rank of the multi-process mailbox all-reduce test (tests/test_gpu_ipc.py starts R of these through
torch.distributed.run; they may all share GPU 0).  Every rank: its contiguous member block of a small ensemble,
collective="ipc", a few evaluations (host and device entry points, L-BFGS), results written to <out>.rank<r>.npz.
Fresh processes only: nothing here replaces a process that has touched the GPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import quoptimalcontrol_jl_amd as qoc
    from quoptimalcontrol_jl_amd.distributed import sharded_engine

    out, cfg, E, N = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    rank = int(os.environ["RANK"])
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")
    w = qoc.workloads.reference_ensemble("StateTransfer", E, N, 5.0) if cfg == "REF" else qoc.workloads.config(cfg, E=E, N=N)
    batch = int(os.environ.get("IPC_TEST_BATCH", "1"))
    sg = sharded_engine(w, dev, collective="ipc", max_batch=batch)
    res = {"collective": np.array(sg.collective), "comm_size": sg.comm_size, "error": np.array(getattr(sg, "attach_error", ""))}
    if sg.collective == "ipc":
        rng = np.random.default_rng(5)
        xs = [w.x] + [w.x + 0.1 * rng.standard_normal(w.x.shape) for _ in range(4)]
        Fs, Gs = [], []
        for x in xs:                                   # (an odd number: both parities of the mailbox slots, twice)
            F, G = sg.eval(x)
            Fs.append(F)
            Gs.append(G)
        res["F"], res["G"] = np.array(Fs), np.array(Gs)
        xd = torch.as_tensor(np.ascontiguousarray(xs[1].T), device=dev)
        fg = sg.eval_device(xd)
        torch.cuda.synchronize(dev)
        res["fg_device"] = fg.cpu().numpy()
        F6, G6 = sg.eval(xs[2])                        # host path right behind the device path
        res["F6"], res["G6"] = F6, G6
        res["names"] = np.array(";".join(sg.local.kernel_names()))
        if batch > 1:                                  # n_x control arrays per call: every rank's rows travel as one exchange
            Fb, Gb = sg.local.eval_batch(np.array(xs[:batch]))
            res["Fb"], res["Gb"] = Fb, Gb
            Fb1, Gb1 = sg.local.eval_batch(np.array(xs[1:2]))      # fewer than max_batch: the whole mailbox still takes part
            res["Fb1"], res["Gb1"] = Fb1, Gb1
        if os.environ.get("IPC_TEST_LBFGS"):
            xm, info = sg.local.lbfgs(w.x, iterations=15)
            res["lbfgs_x"], res["lbfgs_min"], res["lbfgs_evals"] = xm, info["minimum"], info["evaluations"]
    np.savez(f"{out}.rank{rank}.npz", **res)
    dist.barrier()
    sg.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
