"""grape_get_kernel_names (ABI v4): the launch list of the last evaluation, as bench.py labels its lines."""
import numpy as np
import pytest

import quoptimalcontrol_jl_amd as qoc

pytestmark = pytest.mark.gpu


def _names(cfg, **kw):
    w = qoc.workloads.config(cfg, **kw)
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, device=0) as eng:
        assert eng.kernel_names() == []                 # nothing evaluated yet
        eng.eval(w.x)
        first = eng.kernel_names()
        eng.eval(w.x)
        assert eng.kernel_names() == first              # the log is per evaluation, not cumulative
        return first, eng.info


def test_headline_flow_is_the_lane_pair_sweep_and_one_reduction():
    names, info = _names("C3", E=64, N=100)
    assert names[0] == "sweep_pair_kernel" and info["lane_pair"] == 1
    assert all(n.startswith("reduce") for n in names[1:]) and len(names) <= 3


def test_vector_flow_names_follow_grape_info():
    names, info = _names("C4", E=512, N=40)
    assert info["expm_action"] == 1
    assert names[:2] == ["action_rows_kernel", "action_parts_kernel"]
    assert names[2].startswith("action_forms")


def test_small_rank_one_ensemble_takes_the_propagator_chain():
    names, info = _names("C4", E=2, N=40)
    assert info["expm_action"] == 0
    assert any(n.startswith("prop_") for n in names)


def test_bench_line_fits_the_drivers_tail_and_is_labelled_by_the_library():
    """The printed JSON line must stay under 7.5 KB (the driver keeps an 8 KB tail: round 3's 17.8 KB line lost C4 and C2),
    carry both fractions at the top of `roofline`, and name its kernels with what the library launched."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "10", "--warmup", "3", "--cpu-seconds", "1",
                          "--cpu-sample", "4", "--extra-configs", "C4,C2,C4x1"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = out.stdout.strip().splitlines()[-1]
    assert len(line) <= 7680, len(line)
    d = json.loads(line)
    r = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "frac_model_s", "traffic"} <= set(r)
    assert r["kernel"].startswith("sweep_pair_kernel") and 0 < r["frac"] <= 1.0 < 2 * r["frac_model_s"]
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["kind"] == "port"
    ids = [e["id"] for e in d["extra_configs"]]
    assert ids == ["C4", "C2", "C4x1"]
    c4 = d["extra_configs"][0]["roofline"]
    assert "action_parts_kernel" in c4["kernel"] or any("action_parts_kernel" in q["kernel"] for q in c4["parts"])
