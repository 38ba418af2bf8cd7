#!/usr/bin/env python3
"""The gather launch of the bench (100 000 genomes, 4096 queries, K=31 S=15 W=12) under option "gather_variant":
the shipped form against the measurement-only variants of an ABLATION build (make -C niqki_amd/csrc ABLATION=1),
which return wrong counters -- 12: no LDS atomics, 16: look-ups only, 18: no walk, 19: the per-id cost of a one-tile
walk with byte counters on the same lines (PairWalk::apply_model).  One JSON line."""
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
    ap.add_argument("--variants", default="2,19,12,2,19")
    ap.add_argument("--reps", type=int, default=6)
    args = ap.parse_args()
    import torch
    import bench
    import niqki_amd
    K, S, W, H, J = 31, 15, 12, 4, 0.1
    F, N, L, SEED, FAM = 1 << S, args.genomes, 5_000_000, 20261003, 100
    dev = torch.device("cuda", 0)
    t32 = lambda a: torch.from_numpy(np.asarray(a).astype(np.int64)).to(torch.int32).to(dev)  # noqa: E731
    e = niqki_amd.Engine(K=K, S=S, W=W, H=H, J=J, device=0)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.set_option("record_len_hint", L)
    e.reserve(N)
    GB = 256
    seq = torch.zeros(GB * L + niqki_amd.SEQ_PAD, dtype=torch.uint8, device=dev)
    sk = torch.empty((GB, F), dtype=torch.int32, device=dev)
    ro = torch.from_numpy(np.arange(GB + 1, dtype=np.int64) * L).to(dev)
    n_fam = N // FAM
    for g0 in range(0, N, GB):
        n = min(GB, N - g0)
        fam, mem, rate = bench.genome_spec(np.arange(g0, g0 + n), n_fam, FAM)
        e.synth_dev(SEED, t32(fam), t32(mem), t32(rate), n, L, L, seq)
        e.sketch_dev(seq, ro if n == GB else torch.from_numpy(np.arange(n + 1, dtype=np.int64) * L).to(dev), n, sk)
        e.insert_dev(sk, n)
    e.build()
    nq = args.batch
    qsk = torch.empty((nq, F), dtype=torch.int32, device=dev)
    qfam, qmem, qrate = bench.query_spec(np.arange(nq), n_fam)
    for q0 in range(0, nq, GB):
        n = min(GB, nq - q0)
        e.synth_dev(SEED, t32(qfam[q0:q0 + n]), t32(qmem[q0:q0 + n]), t32(qrate[q0:q0 + n]), n, L, L, seq)
        e.sketch_dev(seq, ro if n == GB else torch.from_numpy(np.arange(n + 1, dtype=np.int64) * L).to(dev), n, qsk[q0:q0 + n])
    del seq, sk
    stride = niqki_amd.row_stride(N)
    counts = torch.zeros((nq, stride), dtype=torch.int16, device=dev)
    out = {}
    for v in [int(x) for x in args.variants.split(",")]:
        try:
            e.set_option("gather_variant", v)
        except niqki_amd.NiqkiError:
            out[str(v)] = "not in this build"
            continue
        e.query_counts_dev(qsk, nq, counts, stride)
        e.synchronize()
        e.profile(True)
        e.profile_reset()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            e.query_counts_dev(qsk, nq, counts, stride)
        e.synchronize()
        wall = (time.perf_counter() - t0) / args.reps * 1e3
        ms, n = e.profile_read(niqki_amd.KC_GATHER)
        e.profile(False)
        out.setdefault(str(v), []).append({"gather_class_ms_per_launch": round(ms / max(n, 1), 3), "wall_ms": round(wall, 3)})
    print(json.dumps({"genomes": N, "queries": nq, "tile_genomes": e.tile_genomes(), "by_gather_variant": out}))
    e.close()


if __name__ == "__main__":
    main()
