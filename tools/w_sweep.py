#!/usr/bin/env python3
"""Lane-pair kernel (n = 4): ms per evaluation over the waves per member (0 = the library's own choice) for C3-shaped
ensembles larger / longer than the headline, and for n = 2, 3 (lane-per-chunk kernel) -- the measurement behind the "no more than 16 slices per pair" rule of
grape_create (round 6).   usage: tools/w_sweep.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import quoptimalcontrol_jl_amd as qoc
for E, N in ((2048, 1000), (4096, 1000), (1024, 2000), (4096, 2000), (1024, 1500), (3000, 700)):
    w = qoc.workloads.config("C3", E=E, N=N)
    for W in (0, 2, 4, 8):
        try:
            with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, waves_per_member=W) as eng:
                xf = np.ascontiguousarray(w.x.T)
                call = eng.bind_eval(xf, np.empty_like(xf))
                for _ in range(10): call()
                t0 = time.perf_counter()
                for _ in range(60): call()
                ms = (time.perf_counter() - t0) / 60 * 1e3
                print(f"C3 E={E} N={N} W_req={W}: {ms:.4f} ms; S={eng.info['slices_per_lane']} W={eng.info['waves_per_member']} uni={eng.info['unitary_flow']}")
        except Exception as e:
            print("W", W, "failed", repr(e)[:100])

# n = 2, 3 (lane-per-chunk kernel, 64 chunks per wave): random Hermitian UnitaryGate ensembles
rng = np.random.default_rng(1)
def prob(n, E, N, K=2):
    def herm():
        M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)); return (M + M.conj().T) / 2
    A = np.array([herm() for _ in range(E)]); B = np.broadcast_to(np.array([herm() for _ in range(K)]), (E, K, n, n)).copy()
    q, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    Xi = np.broadcast_to(np.eye(n, dtype=complex), (E, n, n)).copy(); Xt = np.broadcast_to(q, (E, n, n)).copy()
    return A, B, Xi, Xt, np.full(E, 1.0 / E), rng.uniform(-1, 1, (K, N))
for n in (2, 3):
    for E, N in ((1024, 1000), (4096, 1000), (1024, 4000), (4096, 4000), (8192, 500)):
        A, B, Xi, Xt, wts, x = prob(n, E, N)
        for W in (0, 1, 2, 4, 8):
            try:
                with qoc.GrapeEngine("UnitaryGate", A, B, Xi, Xt, wts, 2.0, N, waves_per_member=W) as eng:
                    xf = np.ascontiguousarray(x.T); call = eng.bind_eval(xf, np.empty_like(xf))
                    for _ in range(10): call()
                    t0 = time.perf_counter()
                    for _ in range(60): call()
                    ms = (time.perf_counter() - t0) / 60 * 1e3
                    print(f"n={n} E={E} N={N} W_req={W}: {ms:.4f} ms; S={eng.info['slices_per_lane']} W={eng.info['waves_per_member']}")
            except Exception as e:
                print("W", W, "failed", repr(e)[:80])
