#!/usr/bin/env python3
"""configs[4] of BASELINE.json at a reduced read count, for kernel work on the short-read query path: the bench's
10 000-genome index (S=12 W=10) and its device-generated 150-base reads, sketch + query per batch of 65 536, per
kernel class, with the hit-list form (option "hit_lists") on and off and at several list capacities.  Every variant
must return the same bytes.  `bench.py` carries the reported figure; this prints one JSON line per variant."""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1 << 20)
    ap.add_argument("--genomes", type=int, default=10_000)
    ap.add_argument("--min-score", type=int, default=2)
    ap.add_argument("--ahead", action="store_true", help="also: without the per-class timers, serial and with the two-halves calls")
    ap.add_argument("--gather-variants", default="", help="also the hit-list form under these gather_variant launch shapes (3: 512, 4: 256, 5: 128 threads)")
    args = ap.parse_args()
    import torch
    import bench
    import niqki_amd
    if os.environ.get("NIQKI_EXP_LIB"):      # A/B against another build of the library (tools/bin/, never the product's path)
        from niqki_amd import capi
        capi._LIB = os.path.abspath(os.environ["NIQKI_EXP_LIB"])
    K, S, W, H = 31, 12, 10, 4
    F, N, L, RL, RB, NR = 1 << S, args.genomes, 5_000_000, 150, 65536, args.reads
    dev = torch.device("cuda", 0)
    t32 = lambda a: torch.from_numpy(np.asarray(a).astype(np.int64)).to(torch.int32).to(dev)  # noqa: E731
    t64 = lambda a: torch.from_numpy(np.asarray(a).astype(np.int64)).to(dev)  # noqa: E731
    e = niqki_amd.Engine(K=K, S=S, W=W, H=H, J=0.1, device=0)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.set_option("record_len_hint", L)
    GB = 250
    seq = torch.zeros(GB * L + niqki_amd.SEQ_PAD, dtype=torch.uint8, device=dev)
    ro = t64(np.arange(GB + 1, dtype=np.int64) * L)
    skb = torch.empty((GB, F), dtype=torch.int32, device=dev)
    seed = 20261003 + 2
    for g0 in range(0, N, GB):
        fam, mem, rate = bench.genome_spec(np.arange(g0, g0 + GB), N // 100, 100)
        e.synth_dev(seed, t32(fam), t32(mem), t32(rate), GB, L, L, seq)
        e.sketch_dev(seq, ro, GB, skb)
        e.insert_dev(skb, GB)
    e.build()
    del seq, skb
    rng = np.random.default_rng(20261003)
    src_g = rng.integers(0, N, NR)
    src_off = rng.integers(0, L - RL, NR).astype(np.uint64)
    reads = torch.zeros(NR * RL + niqki_amd.SEQ_PAD, dtype=torch.uint8, device=dev)
    fam, mem, rate = bench.genome_spec(src_g, N // 100, 100)
    e.synth_reads_dev(seed, t32(fam), t32(mem), t32(rate), t64(src_off), t32(np.arange(NR)), 164, NR, RL, RL, reads)
    e.set_option("record_len_hint", RL)
    e.set_option("min_score", args.min_score)
    rro = t64(np.arange(RB + 1, dtype=np.int64) * RL)
    rsk = torch.empty((RB, F), dtype=torch.int32, device=dev)
    rcap = RB * 256
    rho = torch.zeros(RB + 1, dtype=torch.int64, device=dev)
    rhc = torch.zeros(rcap, dtype=torch.int32, device=dev)
    rhg = torch.zeros(rcap, dtype=torch.int32, device=dev)
    ref = None
    forms = [("counter rows", {"hit_lists": 0}), ("hit lists, cap 256", {"hit_lists": 1, "hit_list_cap": 256}),
             ("hit lists, cap 64", {"hit_list_cap": 64}), ("hit lists, cap 512", {"hit_list_cap": 512}),
             ("hit lists, cap 1024", {"hit_list_cap": 1024})]
    forms += [("hit lists, cap 256, gather_variant %s" % v, {"hit_list_cap": 256, "gather_variant": int(v)}) for v in args.gather_variants.split(",") if v]
    if args.ahead:   # the last batch's hits are those of the serial forms: the same bytes are expected
        forms += [("hit lists, cap 256, no profile", {"hit_list_cap": 256, "_profile": 0}),
                  ("hit lists, cap 256, batch i + 1 sketched beside batch i's query (niqki_sketch_ahead / niqki_query_ahead)",
                   {"hit_list_cap": 256, "_ahead": 1, "_profile": 0})]
    for name, opts in forms:
        for k, v in opts.items():
            if not k.startswith("_"):
                e.set_option(k, v)

        def run():
            if opts.get("_ahead"):
                e.sketch_ahead_dev(reads, rro, RB)
                for a in range(0, NR, RB):
                    if a + RB < NR:
                        e.sketch_ahead_dev(reads[(a + RB) * RL:], rro, RB)
                    e.query_ahead_dev(rho, rhc, rhg, rcap)
                return
            for a in range(0, NR, RB):
                e.sketch_dev(reads[a * RL:], rro, RB, rsk)
                e.query_dev(rsk, RB, rho, rhc, rhg, rcap)
        run()
        e.synchronize()
        e.profile(bool(opts.get("_profile", 1)))
        e.profile_reset()
        t0 = time.perf_counter()
        run()
        e.synchronize()
        dt = time.perf_counter() - t0
        prof = {n_: round(e.profile_read(kc)[0], 2) for n_, kc in (("sketch", niqki_amd.KC_SKETCH), ("gather", niqki_amd.KC_GATHER), ("hits", niqki_amd.KC_HITS))}
        e.profile(False)
        got = (rho.cpu().numpy().copy(), rhc.cpu().numpy()[:int(rho[RB])].copy(), rhg.cpu().numpy()[:int(rho[RB])].copy())
        if ref is None:
            ref = got
        same = all(np.array_equal(x, y) for x, y in zip(ref, got))
        per = np.diff(got[0])
        print(json.dumps({"variant": name, "reads_per_s": NR / dt, "ms_per_batch": {k: round(v / (NR // RB), 3) for k, v in prof.items()},
                          "hits_form": e.stat("last_hits_form"), "hits_crc32": zlib.crc32(got[2].tobytes(), zlib.crc32(got[1].tobytes(), zlib.crc32(got[0].tobytes()))), "same_bytes_as_counter_rows": bool(same),
                          "hits_per_read_last_batch": float(per.mean()), "max": int(per.max()),
                          "frac_reads_over": {c: float((per > c).mean()) for c in (64, 256, 512, 1024)}}))
    e.close()


if __name__ == "__main__":
    main()
