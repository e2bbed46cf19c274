#!/usr/bin/env python3
"""bench.py -- GRAPE gradient-evals/sec on the headline config (BASELINE.json):
2-qubit UnitaryGate, 4x4, 4 controls, 500 slices, 1024-member ensemble (SURVEY.md 8d "C3").

One step = one call of the reference's ensemble closure topt(F, G, x) (src/solve.jl:164-196):
all member evaluations + the weighted reduction, with x already resident in HBM and [G, F]
left in HBM (grape_eval_device).  N > 1 GPUs: one process per GPU (torchrun), the ensemble is
sharded in contiguous member blocks and ONE all-reduce of K*N+1 doubles per step completes it.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
         --master-port 29500 bench.py --gpus 8 --steps 200 --warmup 20

Prints ONE JSON line on rank 0 (contract: see the task statement), including
  roofline      dominant kernel (sweep) priced against the 8 TB/s HBM peak with the
                ALGORITHMIC bytes of BASELINE.md's model S, duration from HIP events recorded
                around every sweep launch inside the timed region (on the launch stream);
  cpu_baseline  the C oracle (a port of the reference's serial algorithm) timed on this
                host on a bounded member sample, rank 0 / N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # MI355X FP64 vector = matrix spec; 72 measured (tools/ubench)


def cpu_baseline(workload, seconds, sample_members):
    """Time the oracle (kind 'port', 1 core) on `sample_members` members of the same workload
    and scale to whole-ensemble evaluations per second."""
    import numpy as np
    from oracle import grape_oracle

    w = workload.members(0, min(sample_members, workload.E))
    args = (w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T)
    grape_oracle.ensemble_eval(*args)                      # warm (page in, build)
    n, t0 = 0, time.perf_counter()
    while True:
        F, G, foms, grads = grape_oracle.ensemble_eval(*args, per_member=True)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    member_evals_per_s = n * w.E / el
    # B2 of BASELINE.md: the same port with OpenMP over members on every host core (bounded, ~1/3 of the time)
    try:
        cores = len(os.sched_getaffinity(0))          # cores this process may actually use
    except AttributeError:
        cores = os.cpu_count() or 1
    try:                                              # a cgroup CPU quota caps what affinity shows
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = min(cores, max(1, int(-(-int(q) // int(per)))))
    except Exception:
        pass
    cores = max(1, min(cores, 64))
    wide = workload.members(0, min(max(sample_members, 16 * cores), workload.E))
    wargs = (wide.sys_type, wide.A, wide.B, wide.Xi, wide.Xt, wide.wts, wide.x, wide.T)
    grape_oracle.ensemble_eval(*wargs, n_threads=cores)
    n2, t1 = 0, time.perf_counter()
    while True:
        grape_oracle.ensemble_eval(*wargs, n_threads=cores)
        n2 += 1
        el2 = time.perf_counter() - t1
        if el2 >= seconds / 3:
            break
    return {
        "value": member_evals_per_s / workload.E, "unit": "gradient-evals/s", "cores": 1, "kind": "port",
        "sample": f"{n} x {w.E} of {workload.E} members, {el:.1f} s of oracle/grape_oracle.c (serial, like the "
                  f"reference's member loop); scaled by members",
        "member_evals_per_s": member_evals_per_s,
        "all_cores": {"value": n2 * wide.E / el2 / workload.E, "cores": cores,
                      "sample": f"{n2} x {wide.E} members, OpenMP over members"},
    }, (foms, grads)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", default="C3", help="workload (C2, C3; others once their kernels exist)")
    ap.add_argument("--ensemble", type=int, default=0, help="override E (default: the config's)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong: the config's ensemble is split over the GPUs (north star); "
                         "weak: every GPU gets a full-size ensemble shard")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-sample", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for plumbing tests)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (nccl) and run the all-reduce even with one rank")
    ap.add_argument("--force-general", action="store_true",
                    help="use the general data flow (forward states stored) even for Hermitian generators")
    ap.add_argument("--slices-per-lane", type=int, default=0)
    ap.add_argument("--waves-per-member", type=int, default=0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import quoptimalcontrol_jl_amd as qoc
    from quoptimalcontrol_jl_amd.distributed import sharded_engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    dev_index = local_rank % torch.cuda.device_count()      # (> 1 rank per GPU only in plumbing tests)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    base = qoc.workloads.config(args.config)
    E_cfg = args.ensemble or base.E
    E_total = E_cfg * (world if args.scaling == "weak" else 1)
    w = qoc.workloads.config(args.config, E=E_total) if E_total != base.E else base

    sg = sharded_engine(w, device, force_collective=args.force_dist,
                        flags=qoc.engine.FLAG_TIME_KERNELS | (qoc.engine.FLAG_FORCE_GENERAL if args.force_general else 0),
                        slices_per_lane=args.slices_per_lane, waves_per_member=args.waves_per_member)
    x_dev = torch.as_tensor(np.ascontiguousarray(w.x.T), device=device)     # (K,N) col-major in HBM

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        sg.eval_device(x_dev)
    barrier()
    if sg.local is not None:
        sg.local.kernel_time(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sg.eval_device(x_dev)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kern_ms, kern_n = sg.local.kernel_time() if sg.local is not None else (0.0, 0)
    fg = sg.fg.cpu().numpy()
    info = sg.local.info if sg.local is not None else {}

    if rank == 0:
        evals_per_s = args.steps / elapsed
        # units: with weak scaling the job evaluates world x the config's ensemble per step;
        # report in the metric's unit (evaluations of the CONFIG's E-member ensemble).
        value = evals_per_s * (E_total / E_cfg)
        local = w.members(sg.lo, sg.hi)
        alg_bytes = local.algorithmic_bytes
        avg_ms = kern_ms / max(kern_n, 1)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"{args.config}_E{local.E}")
            except Exception:
                traffic = None
        if info.get("kernel_family") == 1:            # n > 4: FP64 matrix-core kernels, compute-bound
            tf = local.algorithmic_flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
            roof = {"bound": "mfma", "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": tf / FP64_PEAK_TFLOPS, "traffic": traffic,
                    "kernel": "prop_tile_kernel + chain_tile_kernel", "kernel_avg_us": 1e3 * avg_ms,
                    "kernel_launches": kern_n, "algorithmic_flops_per_launch": local.algorithmic_flops}
        else:
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "kernel": "sweep_small_kernel", "kernel_avg_us": 1e3 * avg_ms,
                    "kernel_launches": kern_n, "algorithmic_bytes_per_launch": alg_bytes}
        out = {
            "metric": "GRAPE gradient-evals/sec", "value": value, "unit": "gradient-evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: {w.sys_type} {w.n}x{w.n}, K={w.K}, N={w.N} slices, "
                                   f"ensemble E={E_total} ({'sharded' if world > 1 else 'one GPU'}, "
                                   f"{local.E} members/GPU), T={w.T}",
                       "parallelism": f"ensemble-shard x{world}, one all-reduce of {w.K * w.N + 1} f64 per step"
                                      if world > 1 else "single GPU",
                       "slices_per_lane": info.get("slices_per_lane"),
                       "waves_per_member": info.get("waves_per_member")},
            "member_evals_per_s": evals_per_s * E_total,
            "roofline": roof,
            "F": float(fg[-1]),
        }
        if world == 1 and not args.no_cpu_baseline:
            cb, (foms_ref, grads_ref) = cpu_baseline(w, args.cpu_seconds, args.cpu_sample)
            out["cpu_baseline"] = cb
            m = len(foms_ref)
            ws = w.members(0, m)                       # same members through the HIP path, rows kept
            with qoc.GrapeEngine(ws.sys_type, ws.A, ws.B, ws.Xi, ws.Xt, ws.wts, ws.T, ws.N, device=dev_index,
                                 member_results=True) as chk:
                chk.eval(ws.x)
                foms, grads = chk.member_results()
            gerr = float(np.abs(grads[:m] - grads_ref).max() / np.abs(grads_ref).max())
            ftol = 1e-10 * np.maximum(np.abs(foms_ref), 1e-3 * w.n * w.n)
            ferr = float((np.abs(foms[:m] - foms_ref) / ftol).max())
            out["parity"] = {"members_checked": m, "max_rel_G": gerr, "max_F_err_over_tol": ferr,
                             "tol": 1e-10, "ok": bool(gerr <= 1e-10 and ferr <= 1.0)}
    else:
        out = None

    sg.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    if out is not None:                       # the JSON line is the last thing on stdout
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
