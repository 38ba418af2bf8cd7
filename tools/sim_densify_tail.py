#!/usr/bin/env python3
"""Model of the short-read kernel's closed-form densification tail (nq_sketch.hip, densify_tail) against a plain
simulation of the passes (src/niqki_index.cpp:313-331 in its pass-parallel form): random reads are densified by
windows of 8 passes until at most 16 cells are empty, then once by the passes to the end and once in closed form with
the kernel's rule for ties (a cell that several entries reach in its pass goes to the smallest index at the START of
the tail; the read is handed back when a losing entry has won a cell in an earlier pass of the tail).  Prints how
many reads had a tail, how many were handed back, and the mismatches (must be 0).  CPU only; uses the oracle's hashes.

    python tools/sim_densify_tail.py [S] [W] [reads]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po
S=int(sys.argv[1]) if len(sys.argv)>1 else 12
W=int(sys.argv[2]) if len(sys.argv)>2 else 10
NR=int(sys.argv[3]) if len(sys.argv)>3 else 200
p=po.make_params(31,S,W,4,0.1)
rng=np.random.default_rng(7)
F=1<<S; Fm=F-1
def passes(cells, ent, start, stop_at):
    """pass-parallel process from pass number `start`; ent: list of dict(A,B,v,mk). returns pass count done"""
    s=start
    idle=0
    while True:
        empty=int((cells<0).sum())
        if empty<=stop_at or idle>=F: return s
        prop={}
        for e in ent:
            t=(e['A']+s*e['B'])&Fm
            if cells[t]<0:
                if t not in prop or e['mk']<prop[t]['mk']: prop[t]=e
        for t,e in prop.items():
            cells[t]=e['v']; e['mk']=min(e['mk'],t)
        idle = 0 if prop else idle+1
        s+=1
def inv_odd(o):
    x=o
    for _ in range(3): x=(x*(2-o*x)) & 0xFFFFFFFF
    return x
nb=0; nt=0; bad_exact=0; tailed=0
for r in range(NR):
    L=int(rng.integers(60,300))
    seq=np.frombuffer(b"ACGT",np.uint8)[rng.integers(0,4,L)].copy()
    sk=po.sketch_accumulate(p,seq).astype(np.int64)
    occ=np.nonzero(sk>=0)[0]
    if len(occ)==0 or len(occ)==F: continue
    ent=[dict(A=po.unrev64(int(sk[i]))&0xFFFFFFFF,B=po.rev64(int(sk[i]))&0xFFFFFFFF,v=int(sk[i]),mk=int(i)) for i in occ]
    cells=sk.copy()
    # run windows of 8 passes until empty<=16 (check only at window ends, as the kernel)
    s=0
    while True:
        # do 8 passes
        for _ in range(8):
            prop={}
            for e in ent:
                t=(e['A']+s*e['B'])&Fm
                if cells[t]<0 and (t not in prop or e['mk']<prop[t]['mk']): prop[t]=e
            for t,e in prop.items():
                cells[t]=e['v']; e['mk']=min(e['mk'],t)
            s+=1
        if (cells<0).sum()<=16 or s>F*2: break
    if (cells<0).sum()==0: continue
    tailed+=1
    # reference continuation
    ref=cells.copy(); ent_ref=[dict(e) for e in ent]
    passes(ref,ent_ref,s,0)
    # closed form
    empt=np.nonzero(cells<0)[0]
    res={}
    win_min={id(e):1<<60 for e in ent}; lose_max={id(e):-1 for e in ent}
    for c in empt:
        keys=[]
        for e in ent:
            T=(e['A']+s*e['B'])&0xFFFFFFFF
            b=e['B']&Fm
            j=((b|F)&-(b|F)).bit_length()-1
            odd=((e['B']>>j)|1)&0xFFFFFFFF
            x=(int(c)-T)&0xFFFFFFFF
            if x & ((1<<j)-1): continue
            sv=((x*inv_odd(odd))&Fm)>>j
            keys.append(((sv<<15)|e['mk'],e))
        if not keys: continue
        keys.sort(key=lambda t:t[0])
        wkey,we=keys[0]; sstar=wkey>>15
        res[int(c)]=we['v']; win_min[id(we)]=min(win_min[id(we)],sstar)
        for k,e in keys[1:]:
            if k>>15==sstar: lose_max[id(e)]=max(lose_max[id(e)],sstar); nt+=1
    bail=any(win_min[id(e)]<lose_max[id(e)] for e in ent)
    if bail: nb+=1; continue
    out=cells.copy()
    for c,v in res.items(): out[c]=v
    if not np.array_equal(out,ref): bad_exact+=1
print("S=%d W=%d reads with a tail %d, bails %d, tied losers %d, MISMATCH %d"%(S,W,tailed,nb,nt,bad_exact))
