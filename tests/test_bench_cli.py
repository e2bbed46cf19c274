"""bench.py command line: a multi-GPU request that cannot be honoured must fail, not print n_gpus: 1."""
import os
import subprocess
import sys

from conftest import ROOT


def _run(args, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          env=e, timeout=300)


def test_more_gpus_than_visible_is_an_error():
    import torch
    n = torch.cuda.device_count() + 1
    p = _run(["--gpus", str(max(n, 2)), "--steps", "1", "--warmup", "0"])
    assert p.returncode != 0
    assert "n_gpus" not in p.stdout and "HIP device" in (p.stderr + p.stdout)


def test_world_size_mismatch_is_an_error():
    p = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], env={"WORLD_SIZE": "2", "RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=2" in (p.stderr + p.stdout)


def test_clock_ramp_makes_the_same_number_of_calls_on_every_rank():
    """N > 1: a bench step holds a collective, so the untimed clock ramp must run a count every rank agrees on (from the
    slowest rank's probe time, an all-reduce MAX) -- a loop on each rank's own clock left one rank a call ahead about
    once in ten two-rank runs (mismatched all-reduce, SIGABRT in gloo).  Two "ranks" with different step times here."""
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    counts = []
    for own_step_s in (0.0002, 0.004):                     # a fast and a slow rank; both are told the slow rank's figure
        calls = [0]

        def step():
            calls[0] += 1
            time.sleep(own_step_s)
        n = bench.clock_ramp(step, seconds=0.05, reduce_max=lambda s: 0.005)
        assert n == calls[0]
        counts.append(n)
    assert counts[0] == counts[1] == 10                    # 0.05 s / 0.005 s per step, the 8 probe steps included
    calls = [0]

    def step1():
        calls[0] += 1
        time.sleep(0.001)
    assert bench.clock_ramp(step1, seconds=0.02) == calls[0] >= 5      # one rank: its own clock


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_scaling_forecast_and_compact_line_stay_within_the_drivers_tail():
    """Round 6: the printed line carries a forecast of the 2 / 4 / 8-GPU runs (shard steps timed on one GPU + an all-reduce
    allowance) and a spot-member parity entry per extra config -- and must still fit the 8 KB of stdout the driver keeps."""
    import json
    bench = _bench_module()
    shards = {"C3": {"E512": {"ms_per_step": 0.060}, "E256": {"ms_per_step": 0.049}, "E128": {"ms_per_step": 0.046}},
              "C5": {"E2048": {"ms_per_step": 85.0}, "E1024": {"ms_per_step": 42.5}, "E512": {"ms_per_step": 21.5}},
              "C4": {"error": "boom"}}
    fc = bench.scaling_forecast(shards, allreduce_us=15.0)
    assert fc["allreduce_allowance_us"] == 15.0 and "C4" not in fc
    assert abs(fc["C3"]["2"] - 1.0 / (60e-6 + 15e-6)) < 1e-6 and abs(fc["C3"]["8"] - 1.0 / (46e-6 + 15e-6)) < 1e-6
    assert fc["C5"]["2"] < fc["C5"]["4"] < fc["C5"]["8"]
    roof = {"bound": "hbm", "achieved": 4300.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.54, "frac_model_s": 1.0, "traffic": 2.6e8,
            "kernel": "sweep_pair_kernel;reduce_rows_mf_kernel", "kernel_avg_us": 68.0, "kernel_min_us": 67.0, "kernel_median_us": 68.0,
            "kernel_max_us": 72.0, "kernel_launches": 113, "bytes_per_launch": 296747008,
            "model_s_equivalent": {"algorithmic_bytes_per_launch": 558891008}}
    par = {"members": [0, 341, 682, 1023], "max_rel_G": 2.5e-14, "max_F_err_over_tol": 0.002, "tol": 1e-10, "ok": True}
    extra_ids = ["C4", "C5", "C4dense", "C4expm", "C6", "C7", "C2", "C5x1", "C4x1", "C6x1", "C7x1", "L1d", "C3pm", "C4pm", "C5pm"]
    tile_roof = dict(roof, bound="mfma", unit="TFLOP/s", kernel="ctrl_sum_kernel;grid_prop_kernel",
                     kernels=[{"kernel": "ctrl_sum_kernel;grid_prop_kernel", "avg_us": 85700.0, "bound": "mfma", "frac_mfma": 0.72, "frac_hbm": 0.2},
                              {"kernel": "chain_tile_unitary_kernel;reduce_stage1;reduce_stage2", "avg_us": 83500.0, "bound": "mfma",
                               "frac_mfma": 0.74, "frac_hbm": 0.4}],
                     mfma_utilisation_from_profile={"grid_prop": 0.72, "chain_tile_unitary": 0.746, "source": "x"})
    out = {"metric": "GRAPE gradient-evals/sec", "value": 12000.123, "unit": "gradient-evals/s", "n_gpus": 1, "steps": 20, "warmup": 5,
           "ms_per_step": 0.0833, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "C3: UnitaryGate 4x4, K=4, N=500 slices, ensemble E=1024 (one GPU, 1024 members/GPU), T=2.0; step = x",
                      "parallelism": "single GPU", "collective": None},
           "blocks": {"seconds": [0.00166, 0.00167, 0.00168]}, "roofline": roof,
           "box": {"arch": "gfx950:sramecc+:xnack-", "compute_units": 256, "hbm_gib": 287.98, "clock_ghz": 1.95, "kernel_cycles": {"median": 133000.0}},
           "cpu_baseline": {"value": 1.5, "unit": "gradient-evals/s", "cores": 1, "kind": "port", "sample": "7 x 32 of 1024 members of oracle x",
                            "all_cores": {"value": 22.0, "cores": 16}},
           "parity": {"members_checked": 32, "max_rel_G": 3.6e-14, "max_F_err_over_tol": 0.003, "tol": 1e-10, "ok": True},
           "extra": {"device_resident_pipelined": {"value": 14000.0}, "shard_fixed_overhead": shards["C3"], "scaling_forecast": fc,
                     "lbfgs": [{"problem": "reference n_ens=5", "device_lbfgs": {"iterations": 40, "evaluations": 700, "seconds": 0.02, "minimum": 0.75},
                                "host_driven_scipy": {"iterations": 26, "evaluations": 30, "seconds": 0.01, "minimum": 0.75}}] * 2},
           "extra_configs": [{"id": i, "value": 905.123, "ms_per_step": 1.1049, "steps": 20, "parity": par,
                              "roofline": roof if i in ("C2", "C3pm") else tile_roof} for i in extra_ids],
           "F": 0.123456789}
    for e in out["extra"]["shard_fixed_overhead"].values():
        e["sweep_kernel_us"] = 31.0
    line = json.dumps(bench.compact(out), separators=(",", ":"))
    c = json.loads(line)
    assert len(line) < 7900, len(line)
    assert c["extra"]["scaling_forecast"]["C3"]["8"] > c["extra"]["scaling_forecast"]["C3"]["2"]
    pm = next(e for e in c["extra_configs"] if e["id"] == "C4pm")
    assert pm["vs_shared_controls"] == 1.0 and pm["parity"][0] == 4 and pm["parity"][3] is True
    assert next(e for e in c["extra_configs"] if e["id"] == "C7")["parity"][3] is True
    assert next(e for e in c["extra_configs"] if e["id"] == "C3pm")["vs_shared_controls"] is not None
