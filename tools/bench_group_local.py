#!/usr/bin/env python3
"""All 8 ranks of a slot-sharded 100 000-genome index on ONE GPU (niqki_group with 8 local shards,
device copies standing in for RCCL): what a rank's step costs besides its own sketching -- slice
pack/unpack, shard gather + candidates, candidate look-up, scatter, threshold -- at the weak-scaling
shape of `bench.py --gpus 8` (every rank brings `--per` queries, every shard gathers 8 x per).
Prints per-rank milliseconds per step by kernel class (time of all 8 shards / 8).

    python tools/bench_group_local.py [--genomes 100000] [--per 4096] [--steps 3] [--exchange sparse|dense]
"""
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
    ap.add_argument("--genomes", type=int, default=100000)
    ap.add_argument("--len", type=int, default=5000000)
    ap.add_argument("--per", type=int, default=4096)
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--exchange", default="sparse")
    ap.add_argument("--seed", type=int, default=20240229)
    args = ap.parse_args()
    import torch
    import niqki_amd
    from bench import genome_spec, query_spec
    dev = torch.device("cuda", 0)
    K, S, W, H, J = 31, 15, 12, 4, 0.1
    F = 1 << S
    N, L, G, per = args.genomes, args.len, args.shards, args.per
    n_fam = max(1, N // 100)
    engs = []
    for r in range(G):
        sb, se = niqki_amd.group_slot_range(r, G, S)
        e = niqki_amd.Engine(K=K, S=S, W=W, H=H, J=J, device=0, slot_begin=sb, slot_end=se)
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        e.set_option("record_len_hint", L)
        e.reserve(N)
        engs.append(e)
    grp = niqki_amd.Group(engs)
    grp.set_option("exchange", {"auto": 0, "sparse": 1, "dense": 2}[args.exchange])
    grp.set_option("cand_cap", 256)

    def dev_u32(a):
        return torch.from_numpy(a.astype(np.int64)).to(torch.int32).to(dev)

    GB = 256                                           # genomes per rank and insert round
    seq = torch.zeros(GB * L + niqki_amd.SEQ_PAD, dtype=torch.uint8, device=dev)
    ro = torch.from_numpy(np.arange(GB + 1, dtype=np.int64) * L).to(dev)
    sk = [torch.full((GB, F), -1, dtype=torch.int32, device=dev) for _ in range(G)]
    t0 = time.time()
    for base in range(0, N, G * GB):
        for r in range(G):
            g0 = base + r * GB
            n = max(0, min(GB, N - g0))
            if n:
                fam, mem, rate = genome_spec(np.arange(g0, g0 + n), n_fam, 100)
                engs[0].synth_dev(args.seed, dev_u32(fam), dev_u32(mem), dev_u32(rate), n, L, L, seq)
                engs[0].sketch_dev(seq, ro if n == GB else torch.from_numpy(np.arange(n + 1, dtype=np.int64) * L).to(dev), n, sk[r])
        grp.insert_dev(sk, GB, min(G * GB, N - base))
    for e in engs:
        e.build()
        e.synchronize()
    print("index of %d genomes over %d shards on one GPU: %.1f s" % (N, G, time.time() - t0), file=sys.stderr)
    del seq, sk
    # every rank's queries of one step
    qseq = torch.zeros(per * L + niqki_amd.SEQ_PAD, dtype=torch.uint8, device=dev)
    qro = torch.from_numpy(np.arange(per + 1, dtype=np.int64) * L).to(dev)
    qsk = [torch.empty((per, F), dtype=torch.int32, device=dev) for _ in range(G)]
    for r in range(G):
        fam, mem, rate = query_spec(r * per + np.arange(per), n_fam)
        engs[0].synth_dev(args.seed, dev_u32(fam), dev_u32(mem), dev_u32(rate), per, L, L, qseq)
        engs[0].sketch_dev(qseq, qro, per, qsk[r])
    del qseq
    cap = per * 4096
    off = [torch.zeros(per + 1, dtype=torch.int64, device=dev) for _ in range(G)]
    hc = [torch.zeros(cap, dtype=torch.int32, device=dev) for _ in range(G)]
    hg = [torch.zeros(cap, dtype=torch.int32, device=dev) for _ in range(G)]
    grp.query_dev(qsk, per, off, hc, hg, cap)          # warm-up: sizes the workspaces
    torch.cuda.synchronize()
    for e in engs:
        e.profile(True)
        e.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        grp.query_dev(qsk, per, off, hc, hg, cap)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    names = (("gather", niqki_amd.KC_GATHER), ("hits", niqki_amd.KC_HITS), ("exchange", niqki_amd.KC_EXCHANGE))
    per_rank = {}
    for name, kc in names:
        ms = sum(e.profile_read(kc)[0] for e in engs)
        per_rank[name] = ms / args.steps / G
    hits = int(sum(int(o[per].item()) for o in off))
    out = {"shards_on_one_gpu": G, "index_genomes": N, "queries_per_rank_and_step": per, "exchange": args.exchange,
           "ms_per_step_all_shards": dt * 1e3, "ms_per_step_per_rank": dt * 1e3 / G,
           "per_rank_ms_by_class": per_rank, "hits_per_step": hits, "overflows": grp.stat("overflows"),
           "note": "device copies stand in for the RCCL transfers; a rank's own sketching (bench.py: ~22 ms per 4096 genomes) is not included"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
