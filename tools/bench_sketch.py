#!/usr/bin/env python3
"""Sketch kernel alone: N synthetic 5 Mbp genomes resident in HBM, sketched R times.

    python tools/bench_sketch.py [--n 1024] [--len 5000000] [--reps 5] [--check 2]

Prints G k-mers/s of `niqki_sketch` (HIP events of the library's profile spans) and checks the first
--check sketches against the oracle."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--len", type=int, default=5_000_000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--check", type=int, default=2)
    ap.add_argument("--S", type=int, default=15)
    args = ap.parse_args()
    import torch
    import niqki_amd
    dev = torch.device("cuda", 0)
    K, S, W, H = 31, args.S, 12, 4
    F = 1 << S
    eng = niqki_amd.Engine(K=K, S=S, W=W, H=H, J=0.1, device=0)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_option("record_len_hint", args.len)
    n, L = args.n, args.len
    seq = torch.zeros(n * L + niqki_amd.SEQ_PAD, dtype=torch.uint8, device=dev)
    fam = torch.arange(n, dtype=torch.int32, device=dev) // 4
    mem = torch.arange(n, dtype=torch.int32, device=dev) % 4
    rate = (mem * 100).to(torch.int32)
    eng.synth_dev(7, fam, mem, rate, n, L, L, seq)
    ro = torch.from_numpy(np.arange(n + 1, dtype=np.int64) * L).to(dev)
    sk = torch.empty((n, F), dtype=torch.int32, device=dev)
    eng.sketch_dev(seq, ro, n, sk)
    eng.synchronize()
    eng.profile(True)
    eng.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        eng.sketch_dev(seq, ro, n, sk)
    eng.synchronize()
    dt = time.perf_counter() - t0
    ms, launches = eng.profile_read(niqki_amd.KC_SKETCH)
    kmers = args.reps * n * (L - K)
    print("sketch: %d genomes x %d bp, %d launches: %.3f ms per launch (events), %.1f G k-mers/s (events), %.1f (wall)"
          % (n, L, launches, ms / max(launches, 1), kmers / (ms * 1e-3) / 1e9, kmers / dt / 1e9))
    if args.check:
        from oracle import pyoracle as po
        p = po.make_params(K, S, W, H, 0.1)
        got = sk[:args.check].cpu().numpy()
        for i in range(args.check):
            g = niqki_amd.synth_genome_host(7, i // 4, i % 4, (i % 4) * 100, L)
            assert np.array_equal(got[i], po.compute_sketch(p, g)), "sketch %d differs from the oracle" % i
        print("parity: %d sketches bit-exact vs the oracle" % args.check)


if __name__ == "__main__":
    main()
