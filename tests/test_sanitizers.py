"""CPU sanitizer builds: AddressSanitizer + UndefinedBehaviorSanitizer over
  * the C oracle (oracle/grape_oracle.c) on seeded problems of every shape, and
  * the HOST layer of the C ABI (csrc/grape_api.cpp: argument validation, planning, error paths) built
    with g++ against stand-in kernel launchers (tests/san/stub_kernels.cpp) and driven by a deterministic
    fuzz of grape_config (tests/san/fuzz_abi.cpp).
GPU AddressSanitizer is not available on this pool; device code is covered by the parity tests instead.
A hypothesis fuzz of the same argument space runs against the real libgrape_hip.so through ctypes."""
import ctypes as C
import os
import shutil
import subprocess

import pytest
from hypothesis import given, settings, strategies as st

from conftest import HAS_GPU, ROOT

SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=97", UBSAN_OPTIONS="print_stacktrace=1")


def _build(tmp, name, cmd):
    exe = os.path.join(tmp, name)
    p = subprocess.run(cmd + ["-o", exe], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    return exe


def test_oracle_under_asan_ubsan(tmp_path):
    exe = _build(str(tmp_path), "oracle_san",
                 ["gcc", "-std=c11", "-fopenmp"] + SAN + [os.path.join(ROOT, "oracle", "grape_oracle.c"),
                                                           os.path.join(ROOT, "tests", "san", "oracle_driver.c"), "-lm"])
    p = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=600)
    assert p.returncode == 0 and "oracle sanitizer run ok" in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-3000:])


def test_operator_analysis_under_asan_ubsan(tmp_path):
    """csrc/grape_host.hpp (rank-one factorisation, sparse control lists: what grape_set_operators decides the chain
    kernels from) on 20k random / structured / non-finite inputs with exactly-sized buffers."""
    exe = _build(str(tmp_path), "host_detect_san",
                 ["g++", "-std=c++17", "-I" + os.path.join(ROOT, "quoptimalcontrol.jl_amd", "csrc")] + SAN +
                 [os.path.join(ROOT, "tests", "san", "host_detect.cpp")])
    p = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=600)
    assert p.returncode == 0 and "host detect ok" in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-3000:])


@pytest.mark.skipif(not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"), reason="needs the HIP host headers")
def test_abi_host_layer_fuzz_under_asan_ubsan(tmp_path):
    """grape_api.cpp compiled as plain C++ (host HIP API only) + stub launchers, ASan/UBSan, 20k fuzzed configs.
    Without a GPU every create must fail cleanly (validation / no-device); with one, created contexts are destroyed."""
    src = os.path.join(ROOT, "quoptimalcontrol.jl_amd", "csrc", "grape_api.cpp")
    exe = _build(str(tmp_path), "abi_san",
                 ["g++", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                  "-Wno-deprecated-declarations"] + SAN +
                 [src, os.path.join(ROOT, "tests", "san", "stub_kernels.cpp"), os.path.join(ROOT, "tests", "san", "fuzz_abi.cpp"),
                  "-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-Wl,-rpath,/opt/rocm/lib"])
    env = dict(ENV, ASAN_OPTIONS=ENV["ASAN_OPTIONS"] + ":detect_leaks=0:protect_shadow_gap=0")   # the HIP runtime itself is not leak-clean
    env["HIP_VISIBLE_DEVICES"] = ""                       # host layer only: keep the GPU (if any) out of an ASan process
    p = subprocess.run([exe, "20000"], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0 and "fuzz ok" in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-3000:])


FIELDS = dict(
    sys_type=st.integers(-2, 4), variant=st.integers(-1, 3),
    n=st.sampled_from([-1, 0, 1, 2, 3, 4, 5, 16, 32, 33, 2 ** 31 - 1]),
    n_controls=st.integers(-2, 64), n_slices=st.sampled_from([-1, 0, 1, 7, 500, 10 ** 6, 2 ** 31 - 1]),
    n_ensemble=st.sampled_from([-1, 0, 1, 5, 1024, 2 ** 31 - 1]),
    duration=st.floats(allow_nan=True, allow_infinity=True, width=64), device=st.integers(-2, 9),
    flags=st.integers(0, 255), slices_per_lane=st.integers(-1, 100), waves_per_member=st.integers(-1, 40),
    expm_squarings=st.integers(-2, 70), max_batch=st.integers(-1, 64), n_state_cols=st.integers(-1, 40),
    n_devices=st.integers(-1, 10))


@pytest.mark.skipif(HAS_GPU, reason="argument-space fuzz of the no-GPU error paths (a GPU run would allocate)")
@settings(max_examples=400, deadline=None)
@given(st.fixed_dictionaries(FIELDS), st.lists(st.integers(-2, 9), min_size=8, max_size=8))
def test_hypothesis_fuzz_of_grape_create(qoc_lib, fields, ids):
    cfg = qoc_lib["GrapeConfig"](**fields)
    for i, d in enumerate(ids):
        cfg.device_ids[i] = d
    h = C.c_void_p(1)
    rc = qoc_lib["lib"].grape_create(C.byref(cfg), C.byref(h))
    assert rc in (-1, -2, -3, -8) and not h.value
    assert qoc_lib["lib"].grape_last_error(None)


@pytest.fixture(scope="module")
def qoc_lib(qoc):
    return {"lib": qoc.load_library(), "GrapeConfig": qoc.engine.GrapeConfig}
