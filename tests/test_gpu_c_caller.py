"""GPU: a plain-C program (tests/c_caller/caller.c, gcc, include/grape_hip.h, -lgrape_hip) drives the library
the way the Julia glue's ccalls do; its printed F and G must match the oracle on the same inputs."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, assert_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,E,variant", [(10, 3, 0), (257, 5, 1), (1000, 1, 0)])
def test_c_caller_matches_oracle(tmp_path, qoc, oracle, N, E, variant):
    exe = str(tmp_path / "caller")
    libdir = os.path.dirname(qoc.engine.library_path())
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_caller", "caller.c"), "-o", exe,
                    "-L", libdir, "-lgrape_hip", f"-Wl,-rpath,{libdir}"], check=True)
    out = subprocess.run([exe, str(N), str(E), str(variant)], check=True, capture_output=True, text=True).stdout
    lines = out.strip().splitlines()
    assert lines[0].startswith(f"abi {qoc.engine.ABI_VERSION} arch gfx950")
    F = float(lines[1].split()[1])
    xs = np.array([float(l.split()[1]) for l in lines[2:]])
    Gs = np.array([float(l.split()[3]) for l in lines[2:]])
    K, n = 2, 2
    x = xs.reshape(N, K).T
    G = Gs.reshape(N, K).T
    Sx = np.array([[0, 0.5], [0.5, 0]], complex)
    Sy = np.array([[0, -0.5j], [0.5j, 0]], complex)
    Sz = np.array([[0.5, 0], [0, -0.5]], complex)
    A = np.array([(1.0 + 0.1 * (k - E // 2)) * Sz for k in range(E)])
    B = np.array([[Sx, Sy]] * E)
    Xi = np.array([np.diag([1.0 + 0j, 0])] * E)
    Xt = np.array([np.diag([0, 1.0 + 0j])] * E)
    F_ref, G_ref = oracle.ensemble_eval("StateTransfer", A, B, Xi, Xt, np.full(E, 1.0 / E), x, 1.0, variant=variant)
    assert_parity(F, G, F_ref, G_ref, n, what=f"C caller N={N} E={E} variant={variant}")
