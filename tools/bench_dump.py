#!/usr/bin/env python3
"""Streaming dump / load at scale: an N-genome index (K=31 S=15 W=12) built from the synthetic
generator, exported in slot groups of ~128 MB (niqki_export_dump_slots), imported into a fresh
handle (niqki_import_slots), and queried on both.  Prints one JSON line."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=100_000)
    ap.add_argument("--len", type=int, default=200_000, help="genome length (sketch content does not matter for the dump)")
    args = ap.parse_args()
    import torch
    import niqki_amd
    dev = torch.device("cuda", 0)
    K, S, W, H = 31, 15, 12, 4
    F = 1 << S
    e = niqki_amd.Engine(K=K, S=S, W=W, H=H, J=0.1, device=0)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    N, L = args.genomes, args.len
    GB = 1024
    seq = torch.zeros(GB * L + 64, dtype=torch.uint8, device=dev)
    sk = torch.empty((GB, F), dtype=torch.int32, device=dev)

    def u32(a):
        return torch.from_numpy(a.astype(np.int64)).to(torch.int32).to(dev)
    t0 = time.time()
    for g0 in range(0, N, GB):
        n = min(GB, N - g0)
        g = np.arange(g0, g0 + n)
        e.synth_dev(5, u32(g // 100), u32(g % 100), u32(np.where(g % 100 == 0, 0, 16 + (g % 100) * 8)), n, L, L, seq)
        e.sketch_dev(seq, torch.from_numpy(np.arange(n + 1, dtype=np.int64) * L).to(dev), n, sk)
        e.insert_dev(sk, n)
    e.build()
    e.synchronize()
    t_build = time.time() - t0
    lib = niqki_amd.lib()
    hdr = np.zeros(24, np.uint8)
    assert lib.niqki_export_dump_header(e.h, hdr.ctypes.data) == 0
    slot_bytes = np.zeros(F + 1, np.uint64)
    t0 = time.time()
    assert lib.niqki_export_dump_layout(e.h, slot_bytes.ctypes.data) == 0
    groups, s0 = [], 0
    while s0 < F:
        s1 = s0 + 1
        while s1 < F and int(slot_bytes[s1 + 1] - slot_bytes[s0]) <= (128 << 20):
            s1 += 1
        groups.append((s0, s1))
        s0 = s1
    parts = []
    for s0, s1 in groups:
        want = int(slot_bytes[s1] - slot_bytes[s0])
        buf = np.empty(want, np.uint8)
        size = C.c_uint64(0)
        assert lib.niqki_export_dump_slots(e.h, s0, s1, buf.ctypes.data, want, C.byref(size)) == 0 and size.value == want
        parts.append(buf)
    t_export = time.time() - t0
    payload = int(slot_bytes[F])
    # import
    t0 = time.time()
    p = niqki_amd.Params(31, 15, 12, 4, 0, 0, 0, 0, 0)
    h = C.c_void_p()
    assert lib.niqki_import_begin(C.byref(p), hdr.ctypes.data, C.byref(h)) == 0
    for (s0, s1), buf in zip(groups, parts):
        used = C.c_uint64(0)
        assert lib.niqki_import_slots(h, s0, s1, buf.ctypes.data, buf.size, C.byref(used)) == 0 and used.value == buf.size
    e2 = niqki_amd.Engine(_handle=h)
    e2.build()
    e2.synchronize()
    t_import = time.time() - t0
    # same answers
    q = e.get_sketches(0, 64)
    a, b = e.query(q), e2.query(q)
    same = all(np.array_equal(x, y) for x, y in zip(a, b)) and e2.n_genomes == N
    size2 = np.zeros(F + 1, np.uint64)
    assert lib.niqki_export_dump_layout(e2.h, size2.ctypes.data) == 0
    same = same and np.array_equal(size2, slot_bytes)
    print(json.dumps({"genomes": N, "payload_GB": round(payload / 1e9, 3), "groups": len(groups),
                      "build_s": round(t_build, 2), "export_s": round(t_export, 2), "import_build_s": round(t_import, 2),
                      "export_GBps": round(payload / t_export / 1e9, 2), "import_GBps": round(payload / t_import / 1e9, 2),
                      "identical_answers": bool(same)}))


if __name__ == "__main__":
    main()
