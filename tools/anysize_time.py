#!/usr/bin/env python3
"""Operator sizes beyond the matrix-core families (n > 64, sweep_any.hip: one workgroup per member, scalar FP64 dot products,
operands from HBM / L2): ms per grape_eval and the FP64 rate it amounts to, next to the C oracle on one core -- the number
VERDICT r5 (Weak #7) found missing.  Shapes: a 7-qubit gate (128 x 128, K = 7, N = 200, E = 64: "C7"), a 4-qubit Liouvillian
(256 x 256), and the sizes the parity tests use (65, 100).   usage: tools/anysize_time.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quoptimalcontrol_jl_amd as qoc  # noqa: E402
from oracle import grape_oracle  # noqa: E402


def problem(n, K, N, E, seed):
    rng = np.random.default_rng(seed)

    def herm():
        M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        return (M + M.conj().T) / (2 * np.sqrt(n))
    A = np.array([herm() for _ in range(E)])
    B = np.broadcast_to(np.array([herm() for _ in range(K)]), (E, K, n, n)).copy()
    q, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    Xi = np.broadcast_to(np.eye(n, dtype=complex), (E, n, n)).copy()
    Xt = np.broadcast_to(q, (E, n, n)).copy()
    return A, B, Xi, Xt, np.full(E, 1.0 / E), rng.uniform(-1, 1, (K, N))


import os as _os
SHAPES = ((65, 3, 50, 64), (100, 4, 50, 64), (128, 7, 200, 64), (256, 4, 20, 16))
if _os.environ.get("GRAPE_ANY_ABL"):
    SHAPES = ((128, 7, 200, 64),)
for n, K, N, E in SHAPES:
    A, B, Xi, Xt, wts, x = problem(n, K, N, E, seed=n)
    T = 1.0
    with qoc.GrapeEngine("UnitaryGate", A, B, Xi, Xt, wts, T, N, flags=qoc.engine.FLAG_TIME_KERNELS, member_results=True) as eng:
        xf = np.ascontiguousarray(x.T)
        g = np.empty_like(xf)
        call = eng.bind_eval(xf, g)
        call()
        eng.kernel_time(reset=True)
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            F = call()
        ms = (time.perf_counter() - t0) / reps * 1e3
        names = eng.kernel_names()
        tot_ms, first_ms = (float(np.median(v)) if len(v) else 0.0 for v in eng.kernel_samples())
        foms, grads = eng.member_results()
    t0 = time.perf_counter()
    F_ref, g_ref = grape_oracle.member_eval("UnitaryGate", A[0], B[0], Xi[0], Xt[0], x, T)
    cpu_s = time.perf_counter() - t0
    gerr = float(np.abs(grads[0] - g_ref).max() / np.abs(g_ref).max())
    # model: Taylor-8 (3 products) + ~2 squarings + 3 chain products per slice, 8 n^3 flops per complex product
    flops = E * N * 8.0 * n ** 3 * 8
    print(f"n={n:4d} K={K} N={N:4d} E={E:3d}: {ms:9.2f} ms per evaluation (propagators {first_ms:.2f} + chain {tot_ms - first_ms:.2f}) = {E * N / ms * 1e3:10.0f} member-slices/s, "
          f"~{flops / ms / 1e9:7.2f} TFLOP/s of FP64 (model: 8 products per slice) | C oracle, 1 core: {cpu_s * E * 1e3:9.0f} ms per "
          f"evaluation ({cpu_s * E * 1e3 / ms:5.1f} x) | member 0 vs oracle: max rel G err {gerr:.1e} | {';'.join(names)}")
