#!/usr/bin/env python3
"""What an insert-then-query flip costs at the north-star shape: 100 000-genome index (random sketches: the
build and the gather do not care where a sketch came from), 256 genomes inserted after a query, then the next
query of 64 sketches -- with the delta segment (default) and with incremental_build = 0 (full rebuild)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import niqki_amd  # noqa: E402

dev = torch.device("cuda")
S, W, N, F = 15, 12, 100_000, 1 << 15
res = {}
for inc in (1, 0):
    e = niqki_amd.Engine(K=31, S=S, W=W, H=4, J=0.1)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.set_option("incremental_build", inc)
    e.reserve(N + 4096)
    g = torch.Generator(device=dev).manual_seed(1)
    for a in range(0, N, 4000):
        e.insert_dev(torch.randint(0, 1 << W, (4000, F), dtype=torch.int32, device=dev, generator=g), 4000)
    q = torch.randint(0, 1 << W, (64, F), dtype=torch.int32, device=dev, generator=g)
    new = torch.randint(0, 1 << W, (256, F), dtype=torch.int32, device=dev, generator=g)
    stride = (N + 4096) & ~1
    cnt = torch.zeros((64, stride), dtype=torch.int16, device=dev)
    e.query_counts_dev(q, 64, cnt, stride)
    e.synchronize()
    ts = []
    for k in range(4):
        e.insert_dev(new, 256)
        e.synchronize()
        t0 = time.perf_counter()
        e.query_counts_dev(q, 64, cnt, stride)
        e.synchronize()
        ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    e.query_counts_dev(q, 64, cnt, stride)
    e.synchronize()
    res["incremental" if inc else "full_rebuild"] = {"flip_query_ms": [round(t * 1e3, 2) for t in ts], "steady_query_ms": round((time.perf_counter() - t0) * 1e3, 2),
                                                     "delta_genomes": e.stat("delta_genomes")}
    e.close()
print(json.dumps(res))
