#!/usr/bin/env python3
"""Scale check of the streamed dump / load path (not a test: minutes of data).
Builds an N-genome index at K=31 S=15 W=12 from synthetic genomes, streams the dump
payload out in slot ranges, streams it into a fresh handle, and compares queries."""
import argparse
import ctypes as C
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=20000)
    ap.add_argument("--len", type=int, default=1_000_000)
    args = ap.parse_args()
    import torch
    import niqki_amd
    dev = torch.device("cuda", 0)
    S, F, N, L = 15, 1 << 15, args.genomes, args.len
    e = niqki_amd.Engine(K=31, S=S, W=12, H=4, J=0.1, device=0)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.set_option("record_len_hint", L)
    t = lambda a: torch.from_numpy(a.astype(np.int64)).to(torch.int32).to(dev)  # noqa: E731
    GB = 500
    seq = torch.zeros(GB * L + 64, dtype=torch.uint8, device=dev)
    sk = torch.empty((GB, F), dtype=torch.int32, device=dev)
    ro = torch.from_numpy(np.arange(GB + 1, dtype=np.int64) * L).to(dev)
    first = None
    for b in range(0, N, GB):
        g = np.arange(b, b + GB)
        e.synth_dev(3, t(g // 50), t(g % 50), t(np.where(g % 50 == 0, 0, 20 * (g % 50))), GB, L, L, seq)
        e.sketch_dev(seq, ro, GB, sk)
        e.insert_dev(sk, GB)
        if first is None:
            first = sk[:8].cpu().numpy()
    e.build()
    e.synchronize()
    L_ = niqki_amd.lib()
    hdr = np.zeros(24, np.uint8)
    L_.niqki_export_dump_header(e.h, hdr.ctypes.data)
    slot_bytes = np.zeros(F + 1, np.uint64)
    t0 = time.time()
    assert L_.niqki_export_dump_layout(e.h, slot_bytes.ctypes.data) == 0
    total = int(slot_bytes[F])
    p = niqki_amd.Params(31, 15, 12, 4, 0, 0, 0, 0, 0)
    h2 = C.c_void_p()
    assert L_.niqki_import_begin(C.byref(p), hdr.ctypes.data, C.byref(h2)) == 0
    md5 = hashlib.md5()
    s0, t_exp, t_imp = 0, 0.0, 0.0
    while s0 < F:
        s1 = s0 + 1
        while s1 < F and int(slot_bytes[s1 + 1] - slot_bytes[s0]) <= (256 << 20):
            s1 += 1
        n = int(slot_bytes[s1] - slot_bytes[s0])
        buf = np.empty(max(n, 1), np.uint8)
        size = C.c_uint64(0)
        ta = time.time()
        assert L_.niqki_export_dump_slots(e.h, s0, s1, buf.ctypes.data, n, C.byref(size)) == 0 and size.value == n
        tb = time.time()
        used = C.c_uint64(0)
        assert L_.niqki_import_slots(h2, s0, s1, buf.ctypes.data, n, C.byref(used)) == 0 and used.value == n
        tc = time.time()
        t_exp += tb - ta
        t_imp += tc - tb
        md5.update(buf[:n].tobytes())
        s0 = s1
    e2 = niqki_amd.Engine(_handle=h2)
    a = e.query_counts(first)
    b = e2.query_counts(first)
    print({"genomes": N, "payload_GB": round(total / 1e9, 3), "export_s": round(t_exp, 2), "import_s": round(t_imp, 2),
           "wall_s": round(time.time() - t0, 2), "payload_md5": md5.hexdigest(), "queries_equal": bool(np.array_equal(a, b)),
           "self_counts": [int(a[i, i]) for i in range(4)]})


if __name__ == "__main__":
    main()
