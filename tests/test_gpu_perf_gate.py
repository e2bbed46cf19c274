"""GPU: performance gate of the headline kernel (VERDICT r4 #1c).  The C3 sweep kernel's cost is held in CYCLES -- the
box-independent figure: boxes of the pool differ by several percent in the clock they sustain under this load, which is what
moved the driver's headline between rounds 3 and 4 while the kernel was the same (profiles/r05_C3_ab.txt).

  * wave cycles: every wave of sweep_pair_kernel stamps the shader-cycle counter at its start and end
    (GRAPE_FLAG_PHASE_STAMPS); the median over the 2048 waves must stay within 3 % of the committed count;
  * kernel cycles: HIP-event time of the launch x the clock measured inside the kernel (cycles / 100 MHz real time) --
    includes launch ramp and tail, and carries the noise of two measurements: within 8 %.
The committed numbers live in profiles/perf_gate.json (measured on the MI355X at the commit that last touched the kernel)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c3_sweep_kernel_cycles_within_the_committed_count(qoc):
    gate = json.load(open(os.path.join(ROOT, "profiles", "perf_gate.json")))["C3_E1024"]
    w = qoc.workloads.config("C3")
    # clock + wave cycles from the stamped build
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, flags=qoc.engine.FLAG_PHASE_STAMPS) as eng:
        clocks, cyc = [], []
        for i in range(300):
            eng.eval(w.x)
            if i >= 200 and i % 10 == 0:
                st = eng.phase_stamps().astype(np.int64)
                real_ns = (st[:, 6] - st[:, 5]) * 10.0
                tot = st[:, 4] - st[:, 0]
                ok = real_ns > 0
                clocks.append(float(np.median(tot[ok] / real_ns[ok])))
                cyc.append(float(np.median(tot[ok])))
    clock_ghz, wave_cycles = float(np.median(clocks)), float(np.median(cyc))
    # kernel time from HIP events on the product build
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N,
                         flags=qoc.engine.FLAG_TIME_KERNELS | qoc.engine.FLAG_TIME_SAMPLED) as eng:
        xf = np.ascontiguousarray(w.x.T)
        call = eng.bind_eval(xf, np.empty_like(xf))
        for _ in range(1500):                                # clock ramp
            call()
        eng.kernel_time(reset=True)
        for _ in range(800):
            call()
        tot_ms, _ = eng.kernel_samples()
    kernel_us = float(np.median(tot_ms) * 1e3)
    kernel_cycles = kernel_us * 1e3 * clock_ghz
    print(f"perf gate: clock {clock_ghz:.3f} GHz, wave cycles {wave_cycles:.0f} (committed {gate['wave_cycles']}), "
          f"kernel {kernel_us:.2f} us = {kernel_cycles:.0f} cycles (committed {gate['kernel_cycles']})")
    assert 1.2 < clock_ghz < 2.6, clock_ghz
    assert wave_cycles <= 1.03 * gate["wave_cycles"], (wave_cycles, gate["wave_cycles"])
    assert kernel_cycles <= 1.08 * gate["kernel_cycles"], (kernel_cycles, gate["kernel_cycles"])
