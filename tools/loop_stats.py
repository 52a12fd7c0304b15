#!/usr/bin/env python3
"""Instruction mix of the largest loop of a kernel in a `hipcc -S` listing (VALU / SALU / LDS / s_waitcnt counts):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -mllvm -amdgpu-kernarg-preload-count=16 -Iinclude -Igwinferno_amd/csrc --cuda-device-only -S -o one.s one.hip
    python tools/loop_stats.py one.s
where one.hip instantiates ONE kernel (template __global__ void gwi::scan_kernel<...>(const KArgs);)."""
import re,sys
from collections import Counter
lines=open(sys.argv[1]).read().split('\n')
labels={}
for i,l in enumerate(lines):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: labels[m.group(1)]=i
best=None
def valu(span): return sum(1 for l in lines[span[0]:span[1]] if l.strip().startswith('v_'))
for i,l in enumerate(lines):
    m=re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)',l) or re.search(r's_branch\s+(\.LBB\d+_\d+)',l)
    if m and m.group(1) in labels and labels[m.group(1)]<i:
        span=(labels[m.group(1)],i)
        if best is None or valu(span)>valu(best): best=span   # the loop with the most vector instructions (the sample loop)
def stats(a,b):
    c=Counter()
    for l in lines[a:b]:
        t=l.strip().split()
        if t and not t[0].startswith(('.',';','//')) and not t[0].endswith(':'): c[t[0]]+=1
    return c
c=stats(*best); tot=stats(0,len(lines))
f=lambda c,p: sum(v for k,v in c.items() if k.startswith(p))
print("loop: total",sum(c.values()),"VALU",f(c,'v_'),"SALU",f(c,'s_'),"ds",f(c,'ds_'),"readlane",c['v_readlane_b32'],"writelane",c['v_writelane_b32'],"waitcnt",c['s_waitcnt'],"| kernel total",sum(tot.values()),"VALU",f(tot,'v_'))
