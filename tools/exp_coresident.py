#!/usr/bin/env python3
"""Experiment (round 6): would the sketch kernel and the gather kernel gain from sharing compute units?

Today they cannot: a sketch workgroup holds 128 KB of u32 cells, a gather workgroup 100 KB of counters, a CU has
160 KB of LDS, so the overlap of niqki_sketch_ahead / niqki_query_ahead only fills the tails of the two launches.
Before anyone rebuilds the sketch kernel around a small LDS footprint, this measures what co-residency would buy
with kernels that exist: the bench's gather launch (100 000-genome index, 4096 queries) on one stream, and on a
second handle's stream the SAME sketch kernel on the same 4096 genomes with S = 12 (16 KB of cells: 42 KB of LDS per
workgroup, the same rolling / hashing / filter work per k-mer) -- alone, and beside each other.  With the stock
library the two cannot share a CU either (92 + 96 registers x 4 waves per SIMD); NIQKI_EXP_LIB names a library whose
sketch kernel is compiled for 8 waves per SIMD (64 registers), and gather_variant 3 is the 512-thread gather shape
(2 waves per SIMD x 104 registers): 256 + 208 registers, 42 + 100 KB of LDS, 6 waves per SIMD -- they fit.

Prints one JSON line per configuration: ms alone, ms beside each other, and their ratio to the sum."""
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
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--side-S", type=int, default=12)
    ap.add_argument("--variants", default="0,3")
    ap.add_argument("--priority", action="store_true", help="the gather handle's stream at the top of the priority range, the side handle's at the bottom")
    args = ap.parse_args()
    import torch
    import bench
    import niqki_amd
    from niqki_amd import capi
    if os.environ.get("NIQKI_EXP_LIB"):
        capi._LIB = os.path.abspath(os.environ["NIQKI_EXP_LIB"])
    K, S, W, H, J = 31, 15, 12, 4, 0.1
    F, N, L, per = 1 << S, args.genomes, 5_000_000, args.batch
    dev = torch.device("cuda", 0)
    t32 = lambda a: torch.from_numpy(np.asarray(a).astype(np.int64)).to(torch.int32).to(dev)  # noqa: E731
    t64 = lambda a: torch.from_numpy(np.asarray(a).astype(np.int64)).to(dev)  # noqa: E731
    eng = niqki_amd.Engine(K=K, S=S, W=W, H=H, J=J, device=0)
    if args.priority:
        eng.set_option("stream_priority", 1)
        torch.cuda.set_stream(torch.cuda.ExternalStream(eng.get_stream(), device=dev))
    else:
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_option("record_len_hint", L)
    eng.reserve(N)
    GB = 256
    seq = torch.zeros(GB * L + niqki_amd.SEQ_PAD, dtype=torch.uint8, device=dev)
    skb = torch.full((GB, F), -1, dtype=torch.int32, device=dev)
    ro = t64(np.arange(GB + 1, dtype=np.int64) * L)
    n_fam = max(1, N // 100)
    for g0 in range(0, N, GB):
        n = min(GB, N - g0)
        fam, mem, rate = bench.genome_spec(np.arange(g0, g0 + n), n_fam, 100)
        eng.synth_dev(20261003, t32(fam), t32(mem), t32(rate), n, L, L, seq)
        eng.sketch_dev(seq, ro if n == GB else t64(np.arange(n + 1, dtype=np.int64) * L), n, skb)
        eng.insert_dev(skb, n)
    eng.build()
    eng.synchronize()
    del seq, skb
    qseq = torch.zeros(per * L + niqki_amd.SEQ_PAD, dtype=torch.uint8, device=dev)
    fam, mem, rate = bench.query_spec(np.arange(per), n_fam)
    eng.synth_dev(20261003, t32(fam), t32(mem), t32(rate), per, L, L, qseq)
    d_ro = t64(np.arange(per + 1, dtype=np.int64) * L)
    qsk = torch.empty((per, F), dtype=torch.int32, device=dev)
    eng.sketch_dev(qseq, d_ro, per, qsk)
    cap = per * 4096
    off = torch.zeros(per + 1, dtype=torch.int64, device=dev)
    hc = torch.zeros(cap, dtype=torch.int32, device=dev)
    hg = torch.zeros(cap, dtype=torch.int32, device=dev)
    # the side handle: the same sketch kernel with a small cell array
    S2 = args.side_S
    side = torch.cuda.Stream(device=dev, priority=0)      # (0 = the lowest priority torch hands out)
    e2 = niqki_amd.Engine(K=K, S=S2, W=W, H=H, J=J, device=0)
    e2.set_stream(side.cuda_stream)
    e2.set_option("record_len_hint", L)
    sk2 = torch.empty((per, 1 << S2), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(ts))

    def query():
        eng.query_dev(qsk, per, off, hc, hg, cap)

    def sketch_full():
        eng.sketch_dev(qseq, d_ro, per, qsk)

    def sketch_side():
        e2.sketch_dev(qseq, d_ro, per, sk2)

    ref = None
    t_full = timed(sketch_full)
    t_side = timed(sketch_side)
    for v in [int(x) for x in args.variants.split(",")]:
        eng.set_option("gather_variant", v)
        t_q = timed(query)
        got = (off.cpu().numpy().copy(), hc[:int(off[per])].cpu().numpy().copy())
        if ref is None:
            ref = got
        same = all(np.array_equal(a, b) for a, b in zip(ref, got))

        def both():
            sketch_side()
            query()

        def both_rev():
            query()
            sketch_side()
        t_b = timed(both)
        t_br = timed(both_rev)
        print(json.dumps({"lib": os.path.basename(capi._LIB), "gather_variant": v, "side_S": S2, "priority_streams": bool(args.priority),
                          "ms_sketch_S15_alone": round(t_full, 3), "ms_sketch_side_alone": round(t_side, 3),
                          "ms_query_alone": round(t_q, 3), "ms_beside_sketch_first": round(t_b, 3),
                          "ms_beside_query_first": round(t_br, 3), "sum": round(t_side + t_q, 3),
                          "beside_over_sum": round(min(t_b, t_br) / (t_side + t_q), 3), "same_hits": bool(same)}), flush=True)
    e2.close()
    eng.close()


if __name__ == "__main__":
    main()
