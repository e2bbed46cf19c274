#!/usr/bin/env python3
"""Static instruction mix of one kernel of sweep_small.hip, split at the phase stamps.
usage: tools/isa_mix.py <substring of the mangled kernel name, e.g. Li4ELi0ELi2>"""
import collections
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "quoptimalcontrol.jl_amd", "csrc", sys.argv[2] if len(sys.argv) > 2 else "sweep_small.hip")
out = "/tmp/isa_mix.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-fast-math",
                       "-ffp-contract=on", "-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
key = sys.argv[1]
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN5grape") and key in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
seg, cur = [], []
for l in lines[start:end]:
    s = l.strip()
    if s.startswith(("s_memtime", "s_memrealtime")):
        seg.append(cur)
        cur = []
    elif l.startswith("\t") and s and not s.startswith((".", ";")):
        cur.append(s.split()[0])
seg.append(cur)
print(lines[start].split(":")[0])
for i, sg in enumerate(seg):
    c = collections.Counter(sg)
    g = lambda *p: sum(v for k, v in c.items() if k.startswith(p))
    if len(sg) > 100:
        print(f"seg{i}: n={len(sg)} fp64={g('v_fma', 'v_mul_f64', 'v_add_f64')} mov64={c['v_mov_b64_e32']} "
              f"acc={g('v_accvgpr')} gload={g('global_load')} gstore={g('global_store')} sload={g('s_load')} "
              f"dsread={g('ds_read')} dswrite={g('ds_write')} bperm={c['ds_bpermute_b32']} waits={c['s_waitcnt']}")
for l in lines[end:end + 80]:
    if any(t in l for t in (".vgpr_count", ".agpr_count", ".sgpr_count", "spill_count", "lds_size")):
        print(l.strip())
