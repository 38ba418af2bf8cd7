#!/usr/bin/env python3
"""The sketch call alone on 16 384 random records of ONE length (150 .. 4000 bases), S=12 W=10: ms per batch and a
checksum of the sketches -- where the one-wavefront kernel hands over to the workgroup kernel (NIQKI_SKETCH_WAVE=0: the
workgroup kernel for every length)."""
import os, sys, json, time, numpy as np
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0, ROOT)
import torch, niqki_amd
dev=torch.device("cuda",0)
e=niqki_amd.Engine(K=31,S=12,W=10,H=4,J=0.1,device=0); e.set_stream(torch.cuda.current_stream().cuda_stream)
rng=np.random.default_rng(2)
for L in (150, 300, 420, 500, 1000, 2000, 4000):
    n=16384
    off=(np.arange(n+1,dtype=np.int64)*L)
    seq=torch.from_numpy(np.frombuffer(b"ACGT",np.uint8)[rng.integers(0,4,n*L)].copy())
    seq=torch.cat([seq, torch.zeros(niqki_amd.SEQ_PAD,dtype=torch.uint8)]).to(dev)
    d_off=torch.from_numpy(off).to(dev); sk=torch.empty((n,4096),dtype=torch.int32,device=dev)
    e.set_option("record_len_hint", L)
    e.sketch_dev(seq,d_off,n,sk); e.synchronize()
    ts=[]
    for _ in range(3):
        t0=time.perf_counter(); e.sketch_dev(seq,d_off,n,sk); e.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
    print(json.dumps({"len":L,"ms_per_16384":round(float(np.median(ts)),3),"checksum":int(sk.to(torch.int64).sum().item())}),flush=True)
