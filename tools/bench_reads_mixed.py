#!/usr/bin/env python3
"""The short-read sketch kernel on a batch of MIXED lengths: 65 536 random reads, a fraction of them 300 bases long
among 150-base ones (average <= 200: the one-wavefront kernel takes its 192-entry list).  A 300-base read has more
occupied cells than that list holds: ms per batch of the sketch call alone, by the long reads' share.
NIQKI_EXP_LIB names another build of the library for an A/B."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import niqki_amd
    if os.environ.get("NIQKI_EXP_LIB"):
        from niqki_amd import capi
        capi._LIB = os.path.abspath(os.environ["NIQKI_EXP_LIB"])
    dev = torch.device("cuda", 0)
    e = niqki_amd.Engine(K=31, S=12, W=10, H=4, J=0.1, device=0)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(1)
    n = 65536
    for share in (0.0, 0.01, 0.05, 0.2):
        lens = np.where(rng.random(n) < share, 300, 150).astype(np.int64)
        off = np.zeros(n + 1, np.int64)
        off[1:] = np.cumsum(lens)
        seq = torch.from_numpy(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(off[-1]))].copy())
        seq = torch.cat([seq, torch.zeros(niqki_amd.SEQ_PAD, dtype=torch.uint8)]).to(dev)
        d_off = torch.from_numpy(off).to(dev)
        sk = torch.empty((n, 4096), dtype=torch.int32, device=dev)
        e.set_option("record_len_hint", int(off[-1] // n))
        e.sketch_dev(seq, d_off, n, sk)
        e.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            e.sketch_dev(seq, d_off, n, sk)
            e.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print(json.dumps({"share_of_300_base_reads": share, "avg_len": float(off[-1] / n), "ms_per_batch": round(float(np.median(ts)), 3),
                          "sketch_checksum": int(sk.to(torch.int64).sum().item())}), flush=True)
    e.close()


if __name__ == "__main__":
    main()
