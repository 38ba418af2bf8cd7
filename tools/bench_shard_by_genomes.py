#!/usr/bin/env python3
"""SURVEY.md 8(e)'s alternative to slot-range shards, emulated on one GPU: shard by GENOME range.  Every GPU then holds
ALL slots of N / G genomes, needs the whole sketch of every query (a 64 KB broadcast per query, no reduce), and answers
every query for its genomes; the hit lists are concatenated.  One GPU plays rank 0 of G in the weak-scaling shape of
`bench.py --gpus G` (every rank brings --batch queries per step): per step it sketches its own --batch queries and runs
counters + threshold + order for all G x --batch queries against its N / G genomes.  The exchange (all-gather of
G x batch x 64 KB of sketches per step) is not part of the figure -- like bench.py --shard-of, which is the slot-range
side of the comparison (profiles/r0X_shard_of_8_weak.json).  Reference side: src/niqki_index.cpp:633-687."""
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
    ap.add_argument("--genomes", type=int, default=100_000)
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--len", type=int, default=5_000_000)
    ap.add_argument("--seed", type=int, default=20261003)
    args = ap.parse_args()
    import torch
    import bench
    import niqki_amd
    K, S, W, H, J = 31, 15, 12, 4, 0.1
    F, N, L, G, per = 1 << S, args.genomes, args.len, args.shards, args.batch
    Ns, nq_all = N // G, G * per
    n_fam = N // 100
    dev = torch.device("cuda", 0)
    t32 = lambda a: torch.from_numpy(np.asarray(a).astype(np.int64)).to(torch.int32).to(dev)  # noqa: E731
    e = niqki_amd.Engine(K=K, S=S, W=W, H=H, J=J, device=0)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.set_option("record_len_hint", L)
    GB = 256
    seq = torch.zeros(per * L + niqki_amd.SEQ_PAD, dtype=torch.uint8, device=dev)
    skb = torch.empty((GB, F), dtype=torch.int32, device=dev)
    ro = torch.from_numpy(np.arange(GB + 1, dtype=np.int64) * L).to(dev)
    for g0 in range(0, Ns, GB):                       # genomes [0, N / G): whole families, as a genome-range shard holds them
        n = min(GB, Ns - g0)
        fam, mem, rate = bench.genome_spec(np.arange(g0, g0 + n), n_fam, 100)
        e.synth_dev(args.seed, t32(fam), t32(mem), t32(rate), n, L, L, seq)
        e.sketch_dev(seq, ro if n == GB else torch.from_numpy(np.arange(n + 1, dtype=np.int64) * L).to(dev), n, skb)
        e.insert_dev(skb, n)
    e.build()
    # the queries of one step: G x per, of which this rank sketches the first `per` inside the step
    allsk = torch.empty((nq_all, F), dtype=torch.int32, device=dev)
    d_ro = torch.from_numpy(np.arange(per + 1, dtype=np.int64) * L).to(dev)
    for r in range(G):
        fam, mem, rate = bench.query_spec(r * per + np.arange(per), n_fam)
        e.synth_dev(args.seed, t32(fam), t32(mem), t32(rate), per, L, L, seq)
        e.sketch_dev(seq, d_ro, per, allsk[r * per:(r + 1) * per])
    fam, mem, rate = bench.query_spec(np.arange(per), n_fam)
    e.synth_dev(args.seed, t32(fam), t32(mem), t32(rate), per, L, L, seq)
    own = torch.empty((per, F), dtype=torch.int32, device=dev)
    cap = nq_all * 256
    off = torch.zeros(nq_all + 1, dtype=torch.int64, device=dev)
    hc, hg = torch.zeros(cap, dtype=torch.int32, device=dev), torch.zeros(cap, dtype=torch.int32, device=dev)

    def step():
        e.sketch_dev(seq, d_ro, per, own)
        e.query_dev(allsk, nq_all, off, hc, hg, cap)
    for _ in range(args.warmup):
        step()
    e.synchronize()
    e.profile(True)
    e.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    e.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    prof = {k: e.profile_read(kc)[0] / args.steps for k, kc in (("sketch", niqki_amd.KC_SKETCH), ("gather", niqki_amd.KC_GATHER), ("hits", niqki_amd.KC_HITS))}
    T = float(e.gathered_dev(allsk[:per], per).sum()) / per
    print(json.dumps({
        "what": "rank 0 of %d GENOME-range shards of a %d-genome index, weak scaling: %d queries sketched, %d queries answered against %d genomes per step"
                % (G, N, per, nq_all, Ns),
        "ms_per_step": dt * 1e3, "kernel_ms_per_step": {k: round(v, 3) for k, v in prof.items()},
        "projected_genomes_per_s_if_exchange_is_free": nq_all / dt,
        "tiles": int(e.stat("tiles")), "tile_genomes": e.tile_genomes(), "gather_form": int(e.stat("last_gather_form")),
        "ids_gathered_per_query_and_shard": T,
        "bucket_lines_per_query_and_shard_at_least": F,
        "note": "every query reads one table entry and (at least) one 128-byte bucket line per slot in EVERY shard: F lines = %.1f MB per query and "
                "shard for %.0f ids (%.1f ids per line), against %.1f MB per query and shard with slot ranges (1 / G of the slots, lines "
                "%.0f %% full).  Exchange not included: all-gather of 64 KB (u16 cells) per query to every rank = %.0f MB per rank and step."
                % (F * 128 / 1e6, T, T / F, 11.2 / G, 61, nq_all * F * 2 / 1e6 * (G - 1) / G),
        "hits_total_last_step": int(off[nq_all].item())}))
    e.close()


if __name__ == "__main__":
    main()
