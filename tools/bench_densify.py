#!/usr/bin/env python3
"""The short-read sketch kernel alone (nq::sketch_reads_kernel: ~120 k-mers, then ~420 densification passes per
150-base read, src/niqki_index.cpp:313-331) on configs[4]'s reads, S=12 W=10: its passes reading their targets a window
ahead (the default) against every entry proposing in every pass (NIQKI_DENSIFY_WINDOW=0, read by the library at every
launch).  Both forms must return the same bytes.  Prints one JSON line."""
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
    ap.add_argument("--reads", type=int, default=1 << 20)
    ap.add_argument("--len", type=int, default=150)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("-S", type=int, default=12)
    ap.add_argument("-W", type=int, default=10)
    args = ap.parse_args()
    import torch
    import bench
    import niqki_amd
    K, S, W, H = 31, args.S, args.W, 4
    F, L, RL, RB, NR = 1 << S, 5_000_000, args.len, 65536, args.reads
    dev = torch.device("cuda", 0)
    t32 = lambda a: torch.from_numpy(np.asarray(a).astype(np.int64)).to(torch.int32).to(dev)  # noqa: E731
    t64 = lambda a: torch.from_numpy(np.asarray(a).astype(np.int64)).to(dev)  # noqa: E731
    e = niqki_amd.Engine(K=K, S=S, W=W, H=H, J=0.1, device=0)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.set_option("record_len_hint", RL)
    rng = np.random.default_rng(20261003)
    N = 1000
    src_g = rng.integers(0, N, NR)
    src_off = rng.integers(0, L - RL, NR).astype(np.uint64)
    reads = torch.zeros(NR * RL + niqki_amd.SEQ_PAD, dtype=torch.uint8, device=dev)
    CH = 1 << 20
    for a in range(0, NR, CH):
        gch = src_g[a:a + CH]
        fam, mem, rate = bench.genome_spec(gch, N // 100, 100)
        e.synth_reads_dev(20261005, t32(fam), t32(mem), t32(rate), t64(src_off[a:a + CH]), t32(np.arange(a, a + len(gch))),
                          164, len(gch), RL, RL, reads[a * RL:])
    rro = t64(np.arange(RB + 1, dtype=np.int64) * RL)
    best, first = {}, None
    for mode in ("0", "1", "0", "1"):
        os.environ["NIQKI_DENSIFY_WINDOW"] = mode
        rsk = torch.empty((RB, F), dtype=torch.int32, device=dev)
        e.sketch_dev(reads, rro, RB, rsk)
        e.synchronize()
        for _ in range(args.reps):
            t0 = time.perf_counter()
            for a in range(0, NR, RB):
                e.sketch_dev(reads[a * RL:], rro, RB, rsk)
            e.synchronize()
            best[mode] = min(best.get(mode, 1e9), time.perf_counter() - t0)
        head = rsk[:4096].cpu().numpy()
        assert first is None or np.array_equal(head, first), "the two forms disagree"
        first = head
    print(json.dumps({"reads": NR, "read_len": RL, "S": S, "W": W,
                      "every_entry_every_pass_ms_per_65536_reads": best["0"] / (NR / RB) * 1e3,
                      "window_ahead_ms_per_65536_reads": best["1"] / (NR / RB) * 1e3,
                      "every_entry_every_pass_reads_per_s": NR / best["0"], "window_ahead_reads_per_s": NR / best["1"],
                      "speedup": best["0"] / best["1"], "same_sketches": True}))


if __name__ == "__main__":
    main()
