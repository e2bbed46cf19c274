import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


HAS_GPU = _has_gpu()


def pytest_collection_modifyitems(config, items):
    if HAS_GPU:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# parity metric of BASELINE.md section 2 / SURVEY.md 8d (north star: <= 1e-10 relative, ComplexF64)
PARITY_RTOL = 1e-10


def assert_parity(F, G, F_ref, G_ref, n, rtol=PARITY_RTOL, what=""):
    F_tol = rtol * max(abs(F_ref), 1e-3 * n * n)
    assert abs(F - F_ref) <= F_tol, f"{what}: F {F} vs {F_ref} (|d|={abs(F - F_ref):.3e} > {F_tol:.3e})"
    gmax = np.abs(G_ref).max()
    err = np.abs(np.asarray(G) - np.asarray(G_ref)).max()
    assert err <= rtol * gmax, f"{what}: |G-G_ref|_inf={err:.3e} > {rtol * gmax:.3e}"


@pytest.fixture(scope="session")
def oracle():
    from oracle import grape_oracle
    grape_oracle.build()
    return grape_oracle


@pytest.fixture(scope="session")
def qoc():
    import quoptimalcontrol_jl_amd
    return quoptimalcontrol_jl_amd
