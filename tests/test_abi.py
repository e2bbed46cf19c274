"""The C-ABI library: loads, exports every symbol include/grape_hip.h declares, validates
arguments, and FAILS LOUDLY without a GPU (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import HAS_GPU, ROOT


def declared_functions():
    hdr = open(os.path.join(ROOT, "include", "grape_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(grape_[a-z_]+)\s*\(", hdr)))


def header_abi_version():
    hdr = open(os.path.join(ROOT, "include", "grape_hip.h")).read()
    return int(re.search(r"#define GRAPE_ABI_VERSION (\d+)", hdr).group(1))


def header_struct(name):
    """[(field, ctype, array_len)] of `typedef struct <name> {...}` in include/grape_hip.h."""
    hdr = open(os.path.join(ROOT, "include", "grape_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), hdr, flags=re.S).group(1)
    consts = {k: int(v) for k, v in re.findall(r"#define (GRAPE_\w+) (\d+)", hdr)}
    out = []
    for ctype, field, arr in re.findall(r"(\w+)\s+(\w+)(?:\[(\w+)\])?;", body):
        out.append((field, ctype, int(consts.get(arr, arr)) if arr else 0))
    return out


C_TYPES = {"int32_t": (C.c_int32, "Int32"), "double": (C.c_double, "Float64"), "uint64_t": (C.c_uint64, "UInt64"),
           "char": (C.c_char, "UInt8")}


@pytest.mark.parametrize("cname,pyname", [("grape_config", "GrapeConfig"), ("grape_info", "GrapeInfo"),
                                          ("grape_comm_id", "GrapeCommId")])
def test_ctypes_mirrors_match_the_header(qoc, cname, pyname):
    """field order, types and array lengths of the ctypes mirrors == the header's structs, and the
    compiler's sizeof agrees with ctypes.sizeof (padding included)."""
    want = header_struct(cname)
    got = getattr(qoc.engine, pyname)._fields_
    assert [f for f, _, _ in want] == [f for f, _ in got]
    for (field, ctype, arr), (_, pyt) in zip(want, got):
        base = C_TYPES[ctype][0]
        assert pyt == (base * arr if arr else base), field
    src = '#include <stdio.h>\n#include "grape_hip.h"\nint main(void){printf("%%zu", sizeof(%s)); return 0;}\n' % cname
    exe = os.path.join(ROOT, "tests", ".sizeof_probe")
    try:
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-x", "c", "-", "-o", exe], input=src.encode(),
                       check=True)
        assert int(subprocess.check_output([exe])) == C.sizeof(getattr(qoc.engine, pyname))
    finally:
        if os.path.exists(exe):
            os.remove(exe)


def test_julia_struct_mirror_matches_the_header():
    """julia/GrapeHIP.jl cannot be executed here (no Julia): at least its GrapeConfig mirror must list
    the header's fields in order with the matching Julia types."""
    jl = open(os.path.join(ROOT, "julia", "GrapeHIP.jl")).read()
    body = re.search(r"struct GrapeConfig\n(.*?)\nend", jl, flags=re.S).group(1)
    got = re.findall(r"^\s*(\w+)::([\w{},]+)", body, flags=re.M)
    want = header_struct("grape_config")
    assert [f for f, _ in got] == [f for f, _, _ in want]
    for (field, jt), (_, ctype, arr) in zip(got, want):
        base = C_TYPES[ctype][1]
        assert jt == (f"NTuple{{{arr},{base}}}" if arr else base), field


def test_header_and_binding_agree(qoc):
    assert declared_functions() == sorted(qoc.engine.EXPORTS)


def test_library_exports_every_declared_symbol(qoc):
    lib = qoc.load_library()
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.grape_abi_version() == qoc.engine.ABI_VERSION == header_abi_version()
    out = subprocess.check_output(["nm", "-D", "--defined-only", qoc.library_path()]).decode()
    exported = set(re.findall(r" T (grape_\w+)", out))
    assert exported == set(declared_functions())


def test_library_carries_gfx950_code_only(qoc):
    blob = open(qoc.library_path(), "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx90a", b"gfx942", b"sm_80", b"nvptx"):
        assert other not in blob
    # built for the pool's XNACK setting (csrc/Makefile TARGETS); an xnack+ object in the tree is refused by the GPU pool
    assert b"gfx950:xnack-" in blob and b"xnack+" not in blob


def test_header_is_plain_c():
    src = '#include "grape_hip.h"\nint main(void){grape_config c; (void)c; return GRAPE_ABI_VERSION - 1;}\n'
    p = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                        "-x", "c", "-", "-fsyntax-only"], input=src.encode(), capture_output=True)
    assert p.returncode == 0, p.stderr.decode()


def test_argument_validation(qoc):
    lib = qoc.load_library()
    h = C.c_void_p()
    assert lib.grape_create(None, C.byref(h)) == -1
    cfg = qoc.engine.GrapeConfig(7, 0, 2, 2, 10, 1, 1.0, -1, 0, 0, 0, -1, 0)
    assert lib.grape_create(C.byref(cfg), C.byref(h)) == -1 and not h.value        # bad sys_type
    assert b"sys_type" in lib.grape_last_error(None)
    cfg = qoc.engine.GrapeConfig(0, 0, 2, 0, 10, 1, 1.0, -1, 0, 0, 0, -1, 0)
    assert lib.grape_create(C.byref(cfg), C.byref(h)) == -1                        # K = 0
    cfg = qoc.engine.GrapeConfig(0, 0, 5000, 2, 10, 1, 1.0, -1, 0, 0, 0, -1, 0)
    assert lib.grape_create(C.byref(cfg), C.byref(h)) == -2                        # n = 5000: beyond the generic kernel's indexing
    assert b"n=5000" in lib.grape_last_error(None)
    cfg = qoc.engine.GrapeConfig(0, 0, 4, 2, 10, 1, 1.0, -1, 0, 0, 0, -1, 0, 5)
    assert lib.grape_create(C.byref(cfg), C.byref(h)) == -1                        # n_state_cols > n
    cfg = qoc.engine.GrapeConfig(1, 0, 4, 2, 10, 1, 1.0, -1, 0, 0, 0, -1, 0, 1)
    assert lib.grape_create(C.byref(cfg), C.byref(h)) == -1                        # m < n needs UnitaryGate
    assert b"UnitaryGate" in lib.grape_last_error(None)
    cfg = qoc.engine.GrapeConfig(0, 0, 4, 2, 10, 1, 1.0, -1, 0, 0, 0, -1, 0, 0, 9)
    assert lib.grape_create(C.byref(cfg), C.byref(h)) == -1                        # n_devices > 8
    cfg = qoc.engine.GrapeConfig(0, 0, 16, 2, 10, 1, 1.0, -1, 4, 0, 0, -1, 0)
    assert lib.grape_create(C.byref(cfg), C.byref(h)) == -2                        # phase stamps: n <= 4 only
    assert lib.grape_destroy(None) == 0
    assert lib.grape_eval(None, None, None, None) == -1
    assert lib.grape_comm_unique_id(None) == -1
    assert lib.grape_comm_attach(None, None, 0, 1) == -1


@pytest.mark.skipif(HAS_GPU, reason="checks the no-GPU behaviour")
def test_no_gpu_means_error_not_fallback(qoc):
    wl = qoc.workloads
    w = wl.config("C1")
    with pytest.raises(qoc.GrapeError) as ei:
        qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N)
    assert ei.value.status == -3                                                   # GRAPE_ERR_NO_DEVICE


def test_product_never_touches_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/."""
    pkg = os.path.join(ROOT, "quoptimalcontrol.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                for line in txt.splitlines():
                    s = line.strip()
                    if s.startswith(("#", "//", "*", '"""')) or "oracle/" in s and ("test" in s or "never" in s):
                        continue
                    assert not re.search(r"(import|from)\s+oracle|grape_oracle|libgrape_oracle", s), (f, s)
