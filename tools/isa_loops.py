#!/usr/bin/env python3
"""Static instruction mix of the LOOPS of one kernel (per iteration: MFMA, other vector, LDS, vector-memory, scalar, barriers) and the
SIMD cost model of tools/ubench/pipe_mix applied to it: 64 cycles per v_mfma_f64_16x16x4 + 4.4 per other vector instruction + 11 per
LDS instruction of the same SIMD.   usage: hipcc -S ... -o k.s file.hip ; tools/isa_loops.py k.s <substring of the mangled kernel name>"""
import re,collections,sys
lines=open(sys.argv[1]).read().split('\n')
key=sys.argv[2]
start=next(i for i,l in enumerate(lines) if l.startswith('_Z') and key in l.split(':')[0] and ':' in l)
end=next(i for i in range(start,len(lines)) if 's_endpgm' in lines[i])
labels={}
for i in range(start,end):
    m=re.match(r'^(\.LBB\d+_\d+):',lines[i])
    if m: labels[m.group(1)]=i
def mix(a,b):
    body=[l.strip().split()[0] for l in lines[a:b] if l.startswith('\t') and l.strip() and not l.strip().startswith(('.',';'))]
    c=collections.Counter(body)
    g=lambda *p: sum(v for k,v in c.items() if k.startswith(p))
    mf=g('v_mfma'); va=g('v_')-mf; ds=g('ds_'); gl=g('global_','buffer_','scratch_'); sa=g('s_')
    return dict(n=len(body),mfma=mf,valu=va,ds=ds,vmem=gl,salu=sa,barrier=c['s_barrier'],model=64*mf+4.4*va+11*ds)
print('total',mix(start,end))
loops=[]
for i in range(start,end):
    m=re.search(r's_cbranch\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)',lines[i])
    if m:
        t=m.group(1) or m.group(2)
        if t in labels and labels[t]<i:
            mm=mix(labels[t],i)
            if mm['mfma']>0: print(t,'lines',labels[t]-start,i-start,mm)
for l in lines[end:end+80]:
    if any(t in l for t in ('.num_vgpr','.num_agpr','spill_count','lds_size','scratch_en','private_seg')):
        print(l.strip()[-90:])
