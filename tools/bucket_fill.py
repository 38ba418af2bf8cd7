#!/usr/bin/env python3
"""How full are the 128-byte bucket lines a bench query reads, and would another way of dealing the genomes to the
counter tiles read fewer?  (the gather launch moves 1.46 x the bytes its layout needs.)

Builds bench.py's index (default: the 100 000 genomes of configs[2]) and one of its query batches, then takes a
sample of (query, slot) pairs and, for each, the set of indexed genomes whose stored fingerprint in that slot equals
the query's -- the bucket Index::query_sketch walks (src/niqki_index.cpp:654-660) -- straight from the stored
sketches.  With the members known, any tiling can be priced without building it: the ids of a bucket that fall into
tile t fill ceil(len_t / 64) lines (every bucket starts on a line of 64 u16 ids; a partial line costs a whole one,
profiles/r02_ubench_sector.txt).  Prints a table for profiles/: the accessed-bucket length histogram of the current
layout per tile (with the fill of the last line by length class), and lines / bytes per query for the alternatives.
The current layout's figure is cross-checked against the library's own count of gathered ids (niqki_query_gathered)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=100_000)
    ap.add_argument("--len", type=int, default=5_000_000)
    ap.add_argument("--queries", type=int, default=64, help="sampled queries of the batch")
    ap.add_argument("--slots", type=int, default=512, help="sampled slots")
    ap.add_argument("--seed", type=int, default=20261003)
    args = ap.parse_args()
    import torch
    import bench
    import niqki_amd
    K, S, W, H, J = 31, 15, 12, 4, 0.1
    F, N, L = 1 << S, args.genomes, args.len
    dev = torch.device("cuda", 0)
    t32 = lambda a: torch.from_numpy(np.asarray(a).astype(np.int64)).to(torch.int32).to(dev)  # noqa: E731
    e = niqki_amd.Engine(K=K, S=S, W=W, H=H, J=J, device=0)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.set_option("record_len_hint", L)
    e.reserve(N)
    GB = 256
    seq = torch.zeros(GB * L + niqki_amd.SEQ_PAD, dtype=torch.uint8, device=dev)
    skb = torch.empty((GB, F), dtype=torch.int32, device=dev)
    n_fam = N // 100
    rng = np.random.default_rng(1)
    slots = np.sort(rng.choice(F, args.slots, replace=False))
    d_slots = torch.from_numpy(slots).to(dev)
    sub = torch.empty((N, args.slots), dtype=torch.int32, device=dev)      # stored fingerprints of the sampled slots
    for g0 in range(0, N, GB):
        n = min(GB, N - g0)
        fam, mem, rate = bench.genome_spec(np.arange(g0, g0 + n), n_fam, 100)
        e.synth_dev(args.seed, t32(fam), t32(mem), t32(rate), n, L, L, seq)
        ro = torch.from_numpy(np.arange(n + 1, dtype=np.int64) * L).to(dev)
        e.sketch_dev(seq, ro, n, skb)
        e.insert_dev(skb, n)
        sub[g0:g0 + n] = skb[:n][:, d_slots]
    e.build()
    # one bench batch of 4096 queries (bench.py's step 0), sketched
    NQ = 4096
    fam, mem, rate = bench.query_spec(np.arange(NQ), n_fam)
    qsk = torch.empty((NQ, F), dtype=torch.int32, device=dev)
    for q0 in range(0, NQ, GB):
        e.synth_dev(args.seed, t32(fam[q0:q0 + GB]), t32(mem[q0:q0 + GB]), t32(rate[q0:q0 + GB]), GB, L, L, seq)
        e.sketch_dev(seq, torch.from_numpy(np.arange(GB + 1, dtype=np.int64) * L).to(dev), GB, qsk[q0:q0 + GB])
    e.synchronize()
    del seq, skb
    T_all = e.gathered_dev(qsk, NQ)                    # ids gathered per query, the library's own count
    qs = np.sort(rng.choice(NQ, args.queries, replace=False))
    g = torch.arange(N, device=dev)
    tiles_now, tile_now = int(e.stat("tiles")), e.tile_genomes()

    def deal(pattern, B):
        """blocks of B consecutive genomes dealt to the tiles in `pattern` order (round-robin)"""
        p = torch.tensor(pattern, device=dev)
        return p[(g // B) % len(pattern)], max(pattern) + 1
    schemes = [("2 tiles, blocks of 32 round-robin (the layout)", deal([0, 1], 32)),
               ("2 tiles, single genomes round-robin", deal([0, 1], 1)),
               ("2 tiles, blocks of 128", deal([0, 1], 128)),
               ("2 tiles, ranges of ids", ((g >= (N + 1) // 2).long(), 2)),
               ("2 tiles 3:2 (60 / 40 percent of the genomes), blocks of 32", deal([0, 0, 0, 1, 1], 32)),
               ("2 tiles 9:7 (56 / 44 percent), blocks of 32", deal([0] * 9 + [1] * 7, 32)),
               ("3 tiles, blocks of 32", deal([0, 1, 2], 32)),
               ("4 tiles, blocks of 32", deal([0, 1, 2, 3], 32)),
               ("1 tile: the bound at this line size (not buildable: 16-bit tile ids, counters in LDS)", (torch.zeros(N, dtype=torch.long, device=dev), 1))]
    res = {name: {"lines": 0, "ids": 0, "accesses": 0, "tiles": nt, "max_tile": int(torch.bincount(tid_, minlength=nt).max())}
           for name, (tid_, nt) in schemes}
    edges = [0, 1, 17, 33, 49, 65, 81, 97, 129, 193, 1 << 30]
    hist = np.zeros((tiles_now, len(edges) - 1), dtype=np.int64)
    hist_ids = np.zeros((tiles_now, len(edges) - 1), dtype=np.int64)
    T_sample = []
    for q in qs:
        fp = qsk[q][d_slots]                            # [slots]
        valid = (fp >= 0) & (fp < (1 << W))             # src/niqki_index.cpp:654
        eq = (sub == fp[None, :]) & valid[None, :]      # [N, slots]: members of the touched buckets
        T_sample.append(float(eq.sum().item()) * F / args.slots)
        for name, (tid_, nt) in schemes:
            r = res[name]
            for t in range(nt):
                ln = (eq & (tid_ == t)[:, None]).sum(0)                         # bucket lengths in tile t
                r["lines"] += int(((ln + 63) // 64).sum().item())
                r["ids"] += int(ln.sum().item())
                r["accesses"] += int((ln > 0).sum().item())
                if name == schemes[0][0]:
                    lnh = ln.cpu().numpy()
                    for b in range(len(edges) - 1):
                        m = (lnh >= edges[b]) & (lnh < edges[b + 1])
                        hist[t, b] += int(m.sum())
                        hist_ids[t, b] += int(lnh[m].sum())
    scale = F / args.slots / len(qs)                    # sample -> per query
    Tq = float(np.mean([float(T_all[q]) for q in qs]))
    out = {"index_genomes": N, "tiles_of_the_built_index": tiles_now, "tile_genomes": tile_now,
           "sample": "%d queries x %d slots of bench.py's first batch" % (len(qs), args.slots),
           "ids_per_query_library_count": Tq, "ids_per_query_from_sample": float(np.mean(T_sample)), "schemes": {}}
    print("# bucket lines read per bench query (configs[2]: %d genomes, S=15 W=12), from the members of %d sampled (query, slot) buckets"
          % (N, len(qs) * args.slots))
    print("# ids gathered per query: %.0f by the library's own count over all slots, %.0f extrapolated from the sample" % (Tq, float(np.mean(T_sample))))
    print("# %-88s %6s %10s %12s %10s %12s %8s" % ("dealing of the genomes to counter tiles", "tiles", "max tile", "lines/query", "ids/line", "MB/query", "vs now"))
    base = None
    for name, _ in schemes:
        r = res[name]
        lines_q = r["lines"] * scale
        ids_q = r["ids"] * scale
        mb = lines_q * 128 / 1e6
        if base is None:
            base = mb
        ok = r["max_tile"] <= 65408
        print("  %-88s %6d %10d %12.0f %10.1f %12.2f %8.3f%s" % (name, r["tiles"], r["max_tile"], lines_q,
                                                                 ids_q / max(lines_q, 1), mb, mb / base, "" if ok else "   (tile too large for u16 ids)"))
        out["schemes"][name] = {"tiles": r["tiles"], "max_tile": r["max_tile"], "lines_per_query": lines_q, "ids_per_line": ids_q / max(lines_q, 1),
                                                   "bucket_MB_per_query": mb, "fits_u16_tile_ids": ok}
    print("# accessed bucket lengths of the layout, per tile (share of a query's non-empty bucket reads; ids/line = fill of 64)")
    print("# %-10s" % "ids" + "".join("%12s" % ("%d-%d" % (edges[b], edges[b + 1] - 1) if edges[b + 1] < 1 << 30 else ">=%d" % edges[b]) for b in range(1, len(edges) - 1)))
    for t in range(tiles_now):
        tot = hist[t, 1:].sum()
        print("  tile %d share" % t + "".join("%12.3f" % (hist[t, b] / max(tot, 1)) for b in range(1, len(edges) - 1)))
        fills = [hist_ids[t, b] / max(hist[t, b], 1) for b in range(1, len(edges) - 1)]
        print("  tile %d ids/read" % t + "".join("%10.1f" % f for f in fills))
    out["length_histogram_edges"] = edges
    out["length_histogram_reads"] = hist.tolist()
    out["length_histogram_ids"] = hist_ids.tolist()
    print(json.dumps(out))
    e.close()


if __name__ == "__main__":
    main()
