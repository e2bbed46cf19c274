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
