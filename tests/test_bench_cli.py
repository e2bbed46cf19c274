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
