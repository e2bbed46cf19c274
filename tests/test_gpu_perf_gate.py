"""GPU: performance gate of the headline kernel (VERDICT r4 #1c).  The C3 sweep kernel's cost is held in CYCLES: boxes of the
pool differ by several percent in the clock they sustain under this load, which is what moved the driver's headline between
rounds 3 and 4 while the kernel was the same (profiles/r05_C3_ab.txt).  Every wave of sweep_pair_kernel stamps the
shader-cycle counter at its phase boundaries (GRAPE_FLAG_PHASE_STAMPS):

  * phase A (H build, expm, chunk product: vector-FP64 issue bound) and phase B (scan) cost a number of cycles that does not
    depend on the clock to first order (phase A also issues the P_t stores: 53.3 k cycles at 1.97 GHz, 54.3 k at 2.33 GHz): the
    median over the 2048 waves must stay within 5 % of the committed count -- 3 % for the kernel + 2 % for the pool's clock
    spread; any change of the instruction stream (register allocation, a larger kernarg struct, a new template argument)
    shows here;
  * phase D reads the stored propagators at HBM's rate: its CYCLE count grows with the clock (26 us are 51 k cycles at 1.97 GHz
    and 57 k at 2.2 GHz), so the whole wave is held to 1.05 x the committed count only after scaling its memory-bound share
    to the committed clock;
  * kernel time by HIP events (product build, warm clock): within 15 % of the committed microseconds -- the gross check (the
    pool's boxes ran this kernel at 69.3 .. 76.5 us in round 5: the clock they sustain under it spans 1.76 .. 2.0 GHz).
Committed numbers: profiles/perf_gate.json (MI355X, the commit that last touched the kernel)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c3_sweep_kernel_cycles_within_the_committed_count(qoc):
    gate = json.load(open(os.path.join(ROOT, "profiles", "perf_gate.json")))["C3_E1024"]
    w = qoc.workloads.config("C3")
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, flags=qoc.engine.FLAG_PHASE_STAMPS) as eng:
        rows = []
        for i in range(1200):                                # (a warm clock: the first hundreds of calls run at the boost clock)
            eng.eval(w.x)
            if i >= 900 and i % 20 == 0:
                st = eng.phase_stamps().astype(np.int64)
                real_ns = (st[:, 6] - st[:, 5]) * 10.0
                ok = real_ns > 0
                d = np.diff(st[:, :5], axis=1)[ok]
                rows.append([np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 3]), np.median((st[:, 4] - st[:, 0])[ok]),
                             np.median((st[:, 4] - st[:, 0])[ok] / real_ns[ok])])
    a_cyc, b_cyc, d_cyc, wave_cyc, clock = (float(v) for v in np.median(np.array(rows), axis=0))
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N,
                         flags=qoc.engine.FLAG_TIME_KERNELS | qoc.engine.FLAG_TIME_SAMPLED) as eng:
        xf = np.ascontiguousarray(w.x.T)
        call = eng.bind_eval(xf, np.empty_like(xf))
        for _ in range(1500):
            call()
        eng.kernel_time(reset=True)
        for _ in range(800):
            call()
        tot_ms, _ = eng.kernel_samples()
    kernel_us = float(np.median(tot_ms) * 1e3)
    d_scaled = d_cyc * gate["clock_ghz"] / clock              # the HBM-bound phase at the committed clock
    wave_scaled = wave_cyc - d_cyc + d_scaled
    print(f"perf gate: clock {clock:.3f} GHz; cycles A {a_cyc:.0f} (committed {gate['phase_a_cycles']}), B {b_cyc:.0f} "
          f"({gate['phase_b_cycles']}), D {d_cyc:.0f} -> {d_scaled:.0f} at {gate['clock_ghz']} GHz ({gate['phase_d_cycles']}), "
          f"wave {wave_cyc:.0f} -> {wave_scaled:.0f} ({gate['wave_cycles']}); kernel {kernel_us:.2f} us ({gate['kernel_us']})")
    assert 1.2 < clock < 2.6, clock
    assert a_cyc <= 1.05 * gate["phase_a_cycles"], (a_cyc, gate["phase_a_cycles"])
    assert b_cyc <= 1.05 * gate["phase_b_cycles"], (b_cyc, gate["phase_b_cycles"])
    assert wave_scaled <= 1.05 * gate["wave_cycles"], (wave_scaled, gate["wave_cycles"])
    assert kernel_us <= 1.15 * gate["kernel_us"], (kernel_us, gate["kernel_us"])
