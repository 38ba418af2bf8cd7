#!/usr/bin/env python3
"""Throughput of the GPU record framing (niqki_stage_raw, nq_ingest.hip) on device-resident
FASTA / FASTQ text: kernel time by the library's own HIP events (class KC_INGEST) and the
wall time of the whole call.  Prints one JSON line."""
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
    from niqki_amd.capi import KC_INGEST
    dev = torch.device("cuda", 0)
    e = niqki_amd.Engine(K=31, S=15, W=12, H=4, J=0.1, device=0)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    res = {}
    # whole mode: 200 genome files of 5 Mbp, 70 columns
    g = niqki_amd.synth_genome_host(3, 0, 0, 0, 5_000_000)
    rows = g[:4_999_960].reshape(-1, 70)
    one = b">genome\n" + np.concatenate([rows, np.full((rows.shape[0], 1), 10, np.uint8)], axis=1).tobytes()
    n = 200
    raw = torch.from_numpy(np.frombuffer(one * n, np.uint8).copy()).to(dev)
    raw = torch.cat([raw, torch.zeros(64, dtype=torch.uint8, device=dev)])
    off = np.arange(n + 1, dtype=np.uint64) * len(one)
    ty = np.full(n, ord("A"), np.uint8)
    e.profile(True)
    for tag, args in (("fasta_genomes", (raw, off, ty, False)),):
        e.stage_raw_dev(*args[:3], lines=args[3])
        e.profile_reset()
        torch.cuda.synchronize()
        t0 = time.time()
        reps = 5
        for _ in range(reps):
            info, _ = e.stage_raw_dev(*args[:3], lines=args[3])
        torch.cuda.synchronize()
        dt = (time.time() - t0) / reps
        ms, cnt = e.profile_read(KC_INGEST)
        res[tag] = {"raw_GB": round(len(one) * n / 1e9, 3), "kernels_ms": round(ms / reps, 3),
                    "kernels_GBps": round(len(one) * n / (ms / reps) / 1e6, 1), "call_ms": round(dt * 1e3, 3),
                    "records": info.n_rec, "seq_bytes": info.seq_bytes}
    # lines mode: 150-base reads, FASTA and FASTQ
    rng = np.random.default_rng(1)
    st = rng.integers(0, 5_000_000 - 150, 60000)
    fa = b"".join(b">read%d\n" % i + bytes(g[s:s + 150]) + b"\n" for i, s in enumerate(st))
    fq = b"".join(b"@read%d\n" % i + bytes(g[s:s + 150]) + b"\n+\n" + b"I" * 150 + b"\n" for i, s in enumerate(st))
    for tag, data, t in (("fasta_reads", fa, "A"), ("fastq_reads", fq, "Q")):
        raw = torch.from_numpy(np.frombuffer(data, np.uint8).copy()).to(dev)
        raw = torch.cat([raw, torch.zeros(64, dtype=torch.uint8, device=dev)])
        off = np.array([0, len(data)], np.uint64)
        ty = np.array([ord(t)], np.uint8)
        e.stage_raw_dev(raw, off, ty, lines=True, max_entries=65536)
        e.profile_reset()
        torch.cuda.synchronize()
        t0 = time.time()
        reps = 10
        for _ in range(reps):
            info, _ = e.stage_raw_dev(raw, off, ty, lines=True, max_entries=65536)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / reps
        ms, cnt = e.profile_read(KC_INGEST)
        res[tag] = {"raw_MB": round(len(data) / 1e6, 2), "kernels_ms": round(ms / reps, 3), "call_ms": round(dt * 1e3, 3),
                    "entries": info.n_entry, "reads_per_s_call": round(info.n_entry / dt)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
