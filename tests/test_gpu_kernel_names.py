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
