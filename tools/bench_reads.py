#!/usr/bin/env python3
"""Short-sequence path (BASELINE.json configs[4]): 150 bp reads vs a 10k-genome
index, K=31 S=12 W=10, --indexlines/--querylines semantics (one sketch per read,
densification dominated).  Prints reads/s for sketch+query (parity of this path is covered by
tests/test_gpu_parity.py).  Not the headline metric; see bench.py for that."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=10000)
    ap.add_argument("--reads", type=int, default=262144)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--len", type=int, default=5_000_000)
    args = ap.parse_args()
    import torch
    import niqki_amd
    K, S, W, H, J = 31, 12, 10, 4, 0.1
    F = 1 << S
    dev = torch.device("cuda", 0)
    eng = niqki_amd.Engine(K=K, S=S, W=W, H=H, J=J, device=0)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    N, L = args.genomes, args.len

    def dev_u32(a):
        return torch.from_numpy(a.astype(np.int64)).to(torch.int32).to(dev)
    GB = 256
    seqbuf = torch.zeros(GB * L + 64, dtype=torch.uint8, device=dev)
    skbuf = torch.empty((GB, F), dtype=torch.int32, device=dev)
    t0 = time.time()
    for g0 in range(0, N, GB):
        n = min(GB, N - g0)
        g = np.arange(g0, g0 + n)
        eng.synth_dev(7, dev_u32(g // 100), dev_u32(g % 100), dev_u32(np.where(g % 100 == 0, 0, 16 + (g % 100) * 8)), n, L, L, seqbuf)
        eng.sketch_dev(seqbuf, torch.from_numpy(np.arange(n + 1, dtype=np.int64) * L).to(dev), n, skbuf)
        eng.insert_dev(skbuf, n)
    eng.build()
    eng.synchronize()
    t_index = time.time() - t0
    # reads: 150-base windows of the first 64 indexed genomes, 1 % substitutions
    rng = np.random.default_rng(5)
    src = [niqki_amd.synth_genome_host(7, g // 100, g % 100, 0 if g % 100 == 0 else 16 + (g % 100) * 8, 200_000) for g in range(64)]
    R = args.reads
    which = rng.integers(0, 64, R)
    start = rng.integers(0, 200_000 - 150, R)
    reads = np.stack([src[w][s:s + 150] for w, s in zip(which[:4096], start[:4096])])
    reads = np.tile(reads, (R // 4096 + 1, 1))[:R].copy()
    sub = rng.random((R, 150)) < 0.01
    reads[sub] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(sub.sum()))]
    d_reads = torch.from_numpy(reads.reshape(-1)).to(dev)
    d_reads = torch.cat([d_reads, torch.zeros(64, dtype=torch.uint8, device=dev)])
    B = args.batch
    ro = torch.from_numpy(np.arange(B + 1, dtype=np.int64) * 150).to(dev)
    sk = torch.empty((B, F), dtype=torch.int32, device=dev)
    cap = B * 64
    hit_off = torch.zeros(B + 1, dtype=torch.int64, device=dev)
    hc = torch.zeros(cap, dtype=torch.int32, device=dev)
    hg = torch.zeros(cap, dtype=torch.int32, device=dev)
    eng.set_option("record_len_hint", 150)
    eng.profile(True)
    times = []
    for it in range(R // B + 1):
        b0 = (it % max(R // B, 1)) * B
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.sketch_dev(d_reads[b0 * 150:], ro, B, sk)
        eng.query_dev(sk, B, hit_off, hc, hg, cap)
        eng.synchronize()
        times.append(time.perf_counter() - t0)
    prof = {k: eng.profile_read(v) for k, v in (("sketch", niqki_amd.KC_SKETCH), ("gather", niqki_amd.KC_GATHER), ("hits", niqki_amd.KC_HITS))}
    best = min(times[1:]) if len(times) > 1 else times[0]
    print(json.dumps({"metric": "reads/s (150 bp, sketch+query), %d-genome index, K=31 S=12 W=10" % N,
                      "value": B / best, "batch": B, "ms_per_batch": best * 1e3, "index_build_s": t_index,
                      "kernels_ms_total": {k: round(v[0], 2) for k, v in prof.items()}, "launches": prof["sketch"][1],
                      "hits_total": int(hit_off[B].item())}))


if __name__ == "__main__":
    main()
