#!/usr/bin/env python3
"""bench.py -- query throughput of the MI355X NIQKI engine on BASELINE.json's
metric: query genomes/sec + achieved HBM GB/s on a 100k-genome index,
K=31 S=15 W=12 (H=4, J=0.1 -> min_score 3276), synthetic 5 Mbp genomes.

    python bench.py [--gpus N --steps K --warmup W]

One "step" = one pass of the hot path (k-mer rolling hash -> HyperMinHash
sketch -> densification -> gather-histogram over the inverted index ->
threshold + ordered hits) over one batch of query genomes whose bases are
already resident in HBM (a ring of distinct batches, step i uses batch i mod ring).  N > 1 (torch.distributed.run, one rank per GPU over
RCCL): the index is sharded by sketch-slot range, the per-genome hit vectors are
summed across ranks by a reduce-scatter (niqki_amd/dist.py); total work is
fixed, so scaling is "strong".

Prints ONE JSON line (rank 0).  `roofline` is for the gather-histogram kernel
(the HBM-bound kernel SURVEY.md 8d grades), timed live with HIP events on the
engine's stream; `kernels` lists every kernel class so the ALU-bound sketch
kernel's share is visible too.  `cpu_baseline` is the oracle (a port of the
reference's CPU path, oracle/niqki_oracle.c) on this host's cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def genome_spec(g, n_fam, fam_size):
    """Indexed genome g: family g // fam_size, member g % fam_size; member 0 is the
    ancestor, the others carry substitution rates spread geometrically over
    0.1 % .. 5 % so that in-family Jaccard spans ~0.05 .. 0.95."""
    fam = g // fam_size
    mem = g % fam_size
    rate = np.where(mem == 0, 0, np.round(16 * (820 / 16) ** ((mem - 1) / max(fam_size - 2, 1)))).astype(np.uint32)
    return fam.astype(np.uint32), mem.astype(np.uint32), rate


def query_spec(q, n_fam):
    """Query q: a fresh mutant (member id >= 2^20) of a pseudo-random indexed
    family; every 10th query comes from a family that is not indexed."""
    h = (q.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(33)
    fam = (h % np.uint64(min(n_fam, int(os.environ.get("NIQKI_BENCH_QFAM") or n_fam)))).astype(np.uint32)
    fam = np.where(q % 10 == 9, n_fam + q, fam).astype(np.uint32)
    mem = ((1 << 20) + q).astype(np.uint32)
    rate = (16 + (h >> np.uint64(8)) % np.uint64(400)).astype(np.uint32)
    return fam, mem, rate


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=9)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genomes", type=int, default=100_000)
    ap.add_argument("--batch", type=int, default=4096, help="query genomes per step (whole job)")
    ap.add_argument("--ring", type=int, default=3,
                    help="distinct query batches kept resident in HBM; step i uses batch i mod ring")
    ap.add_argument("--len", type=int, default=5_000_000)
    ap.add_argument("--family", type=int, default=100)
    ap.add_argument("--seed", type=int, default=20261003)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--exchange", default=os.environ.get("NIQKI_EXCHANGE", "auto"),
                    help="cross-shard sum: auto | sparse | reduce_scatter | all_to_all (niqki_amd/dist.py)")
    args = ap.parse_args()

    # Only the JSON line may reach stdout: libraries (RCCL prints a version banner)
    # get stderr for the whole run, the result is written to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import niqki_amd
    from niqki_amd.dist import ShardedQuery, slot_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        log("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # NIQKI_FORCE_DIST=1 runs the sharded (collective) code path even on one rank
    use_dist = world > 1 or os.environ.get("NIQKI_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    K, S, W, H, J = 31, 15, 12, 4, 0.1
    F = 1 << S
    N, L = args.genomes, args.len
    n_fam = max(1, N // args.family)
    sb, se = slot_range(rank, world, F)
    eng = niqki_amd.Engine(K=K, S=S, W=W, H=H, J=J, device=local_rank, slot_begin=sb, slot_end=se)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_option("record_len_hint", L)
    stride_b = L  # records are stored back to back; NIQKI_SEQ_PAD bytes follow the last one

    def dev_u32(a):
        return torch.from_numpy(a.astype(np.int64)).to(torch.int32).to(dev)

    def rec_offsets(n):
        return torch.from_numpy((np.arange(n + 1, dtype=np.int64) * stride_b)).to(dev)

    # ---- index build (not timed): synth -> sketch -> (all_gather) -> insert ----
    t0 = time.time()
    eng.reserve(N)
    GB = 256
    seqbuf = torch.zeros(GB * stride_b + niqki_amd.SEQ_PAD, dtype=torch.uint8, device=dev)
    skbuf = torch.empty((GB, F), dtype=torch.int32, device=dev)
    ro_full = rec_offsets(GB)

    def sketch_padded(n, out):
        eng.sketch_dev(seqbuf, ro_full if n == GB else rec_offsets(n), n, out)

    n_rounds = ((N + GB - 1) // GB + world - 1) // world
    gath = torch.empty((world, GB, F), dtype=torch.int32, device=dev) if use_dist else None
    for r in range(n_rounds):
        b = r * world + rank
        g0 = b * GB
        n = max(0, min(GB, N - g0))
        if n:
            fam, mem, rate = genome_spec(np.arange(g0, g0 + n), n_fam, args.family)
            eng.synth_dev(args.seed, dev_u32(fam), dev_u32(mem), dev_u32(rate), n, L, stride_b, seqbuf)
            sketch_padded(n, skbuf)
        if use_dist:
            eng.synchronize()
            dist.all_gather_into_tensor(gath.view(-1), skbuf.view(-1))
            torch.cuda.synchronize()
            for rr in range(world):
                gg0 = (r * world + rr) * GB
                nn = max(0, min(GB, N - gg0))
                if nn:
                    eng.insert_dev(gath[rr], nn)
        elif n:
            eng.insert_dev(skbuf, n)
    eng.build()
    eng.synchronize()
    t_index = time.time() - t0
    log("[rank %d] index: %d genomes, tile %d, built in %.1f s" % (rank, eng.n_genomes, eng.tile_genomes(), t_index))
    del seqbuf

    # ---- query inputs resident in HBM: a ring of distinct batches, this rank's share ----
    per = (args.batch + world - 1) // world
    n_steps_all = args.warmup + args.steps
    n_batches = max(1, min(n_steps_all, args.ring))
    qseq = torch.zeros(n_batches * per * stride_b + niqki_amd.SEQ_PAD, dtype=torch.uint8, device=dev)
    for bi in range(n_batches):
        q = bi * per * world + rank * per + np.arange(per)
        fam, mem, rate = query_spec(q, n_fam)
        eng.synth_dev(args.seed, dev_u32(fam), dev_u32(mem), dev_u32(rate), per, L, stride_b,
                      qseq[bi * per * stride_b:])
    d_ro = rec_offsets(per)
    qsk = torch.empty((n_batches, per, F), dtype=torch.int32, device=dev)
    cap = per * 4096
    hit_off = torch.zeros((n_steps_all, per + 1), dtype=torch.int64, device=dev)
    hc = torch.zeros(cap, dtype=torch.int32, device=dev)
    hg = torch.zeros(cap, dtype=torch.int32, device=dev)
    stride = (N + 1) & ~1
    counts = torch.zeros((per * world, stride), dtype=torch.int16, device=dev)
    # 256 candidates per query and shard (a family has 100 members); the calibration pass below
    # switches to the dense exchange if any list overflows
    sq = ShardedQuery(eng, N, F, dev, exchange=args.exchange, cand_cap=256, compact_sketches=True) if use_dist else None
    eng.synchronize()

    def step(si):
        bi = si % n_batches
        base = qseq[bi * per * stride_b:]
        eng.sketch_dev(base, d_ro, per, qsk[bi])
        if use_dist:
            sq.step(qsk[bi], hit_off[si], hc, hg, cap)
        else:
            eng.query_counts_dev(qsk[bi], per, counts, stride)
            eng.hits_from_counts_dev(counts, per, stride, 0, N, hit_off[si], hc, hg, cap)

    def barrier():
        if use_dist:
            dist.barrier()

    if use_dist and sq.exchange == "sparse":
        # exchange calibration (untimed, not a warm-up step): every resident batch once, so that
        # a candidate list that does not fit is seen before anything is timed
        for bi in range(n_batches):
            step(bi)
    for bi in range(args.warmup):
        step(bi)
    eng.synchronize()
    torch.cuda.synchronize()
    if use_dist and sq.exchange == "sparse" and int(sq.overflow.item()):
        # a candidate list did not fit: this workload needs the dense exchange
        log("[rank %d] sparse exchange overflowed its candidate capacity, using reduce_scatter" % rank)
        sq.exchange = "reduce_scatter"
        sq.overflow.zero_()
    eng.profile(True)
    eng.profile_reset()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for si in range(args.warmup, n_steps_all):
        step(si)
    eng.synchronize()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    prof = {name: eng.profile_read(kc) for name, kc in (
        ("sketch", niqki_amd.KC_SKETCH), ("densify", niqki_amd.KC_DENSIFY), ("gather", niqki_amd.KC_GATHER),
        ("hits", niqki_amd.KC_HITS))}
    eng.profile(False)

    # ---- roofline of the gather kernel: algorithmic bytes 4T + 20F per query ----
    f_local = se - sb
    T = 0
    for si in range(args.warmup, n_steps_all):
        bi = si % n_batches
        if use_dist:
            allsk = sq.exchange_sketches(qsk[bi])
            T += int(eng.gathered_dev(allsk, per * world).sum())
        else:
            T += int(eng.gathered_dev(qsk[bi], per).sum())
    n_q_local = args.steps * per * world  # queries this GPU's gather kernel saw
    gather_ms, gather_launches = prof["gather"]
    alg_bytes = 4 * T + 20 * f_local * n_q_local
    achieved = alg_bytes / (gather_ms * 1e-3) / 1e9 if gather_ms > 0 else 0.0
    total_hits = int(hit_off[args.warmup:, per].sum().item())
    overflow = bool((hit_off[:, per] > cap).any().item())

    # HBM bytes of the gather kernel from the committed PMC passes (separate rocprofv3
    # --pmc runs of this same command; bench.py cannot collect counters itself)
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "gather_traffic.json")))
        if (tj["index_genomes"], tj["query_batch"], tj["tile_genomes"]) == (N, per * world, eng.tile_genomes()) and world == 1:
            traffic = tj["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(eng, niqki_amd, qseq, qsk, hit_off, hc, hg, args, stride_b, L, N, per, (K, S, W, H, J))

    # bytes a rank sends per step in the exchange (sketch slices, then the candidate lists or the
    # dense counters): with the step time this bounds the average xGMI rate per GPU
    xbytes = None
    if use_dist:
        g1 = (world - 1) / world
        nq_all = per * world
        xbytes = per * F * 2 * g1                                   # all_to_all of F/G-slot sketch slices (int16)
        if sq.exchange == "sparse":
            xbytes += nq_all * sq.cand_cap * 4 * g1 + nq_all * 4 * g1    # all_gather of candidates + their counts
            xbytes += nq_all * world * sq.cand_cap * 4 * g1              # reduce_scatter of the candidates' partial counts
        else:
            xbytes += nq_all * (stride // 2) * 4 * g1                    # dense u16 counters as int32 pairs
    if rank == 0:
        n_queries = args.steps * per * world
        out = {
            "metric": "query genomes/sec, %dk-genome index, K=31 S=15 W=12" % (N // 1000),
            "value": n_queries / dt,
            "unit": "genomes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": "%d synthetic %d bp genomes indexed (families of %d, 0.1-5%% substitutions), "
                            "%d query genomes per step resident in HBM, K=31 S=15 W=12 H=4 J=0.1"
                            % (N, L, args.family, per * world),
                "index_genomes": N, "query_batch": per * world, "genome_len": L,
                "parallelism": "slot-shard x%d (%s exchange)" % (world, sq.exchange) if use_dist else "1 GPU",
                "exchange_overflow": bool(int(sq.overflow.item())) if use_dist else False,
                "exchange_bytes_per_rank_per_step": xbytes,
                "exchange_avg_gbs_per_rank": (xbytes / (dt / args.steps) / 1e9) if xbytes else None,
                "tile_genomes": eng.tile_genomes(), "index_build_s": round(t_index, 2),
                "hits_per_query": total_hits / max(1, args.steps * per), "hit_overflow": overflow,
            },
            "roofline": {
                "kernel": "nq::gather_kernel (gather-histogram, rank 0's slot shard)",
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes / max(1, gather_launches),
                "launches": gather_launches, "avg_launch_ms": gather_ms / max(1, gather_launches),
                "gathered_ids_per_query": T / max(1, n_q_local),
            },
            "kernels": {k: {"ms": round(v[0], 3), "launches": v[1]} for k, v in prof.items()},
            # the sketch kernel is integer-ALU bound (4 64-bit multiplies per k-mer):
            # its rate in k-mers and the HBM bytes it needs (1 byte per base + the sketch)
            "sketch_kernel": {
                "gkmers_per_s": (args.steps * per * max(L - K, 0)) / (prof["sketch"][0] * 1e-3) / 1e9 if prof["sketch"][0] else 0.0,
                "hbm_gbs": (args.steps * per * (L + 4 * F)) / (prof["sketch"][0] * 1e-3) / 1e9 if prof["sketch"][0] else 0.0,
                "bound": "valu",
            },
            "cpu_baseline": cpu,
        }
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    eng.close()
    if use_dist:
        dist.destroy_process_group()


def cpu_baseline(eng, niqki_amd, qseq, qsk, hit_off, hc, hg, args, stride_b, L, N, per, prm):
    """The oracle (port of the reference CPU path) on this host, on a bounded
    sample of the same workload; also the in-run parity check."""
    from oracle import pyoracle as po
    K, S, W, H, J = prm
    p = po.make_params(K, S, W, H, J)
    cores = po.lib().nqo_max_threads()
    n_s = int(min(per, max(8, 2 * cores)))
    bi = args.warmup % qsk.shape[0]   # the batch the first timed step used
    seqs = qseq[bi * per * stride_b: bi * per * stride_b + n_s * stride_b].cpu().numpy()
    rec = np.stack([seqs[i * stride_b:i * stride_b + L] for i in range(n_s)])
    rec_off = (np.arange(n_s + 1) * L).astype(np.uint64)
    # the host may hand this process fewer CPUs than it has threads: time the sketch leg
    # at a few thread counts and keep the fastest (that count is what `cores` reports)
    t_sk, sk_cpu, best_threads = None, None, cores
    for th in sorted({cores, max(1, cores // 2), max(1, cores // 4)}, reverse=True):
        t0 = time.perf_counter()
        out = po.sketch_batch(p, rec.reshape(-1), rec_off, threads=th)
        t = time.perf_counter() - t0
        if t_sk is None or t < t_sk:
            t_sk, sk_cpu, best_threads = t, out, th
    cores = best_threads
    sk_gpu = qsk[bi, :n_s].cpu().numpy()
    parity_sketch = bool(np.array_equal(sk_cpu, sk_gpu))
    # gather leg on sub-indexes of the first n genomes, extrapolated linearly in N
    pts = []
    parity_counts = True
    for n_sub in (4096, 16384):
        n_sub = min(n_sub, N)
        sub = eng.get_sketches(0, n_sub)
        ix = po.Index(p, sub)
        best = None
        for _ in range(3):  # first pass warms the pages, keep the fastest
            t0 = time.perf_counter()
            off, c, g = ix.query_batch(sk_cpu, threads=cores)
            t = time.perf_counter() - t0
            best = t if best is None else min(best, t)
        pts.append((n_sub, best))
        if n_sub == min(16384, N):
            cnt = eng.query_counts(sk_gpu)[:, :n_sub]
            for i in range(min(n_s, 4)):
                parity_counts &= bool(np.array_equal(cnt[i].astype(np.uint32), ix.counts(sk_cpu[i])))
        del ix
    (n1, t1), (n2, t2) = pts
    if n2 == n1:
        t_q = t2
    elif t2 > t1:
        t_q = t1 + (t2 - t1) * (N - n1) / (n2 - n1)   # linear in N through both points
    else:
        t_q = t2                                      # no measurable growth: take the larger index as is
    val = n_s / (t_sk + t_q)
    return {
        "value": val, "unit": "genomes/s", "cores": cores, "kind": "port",
        "sample": "%d query genomes of step %d: sketch leg timed in full (%.2f s); gather leg timed on "
                  "sub-indexes of %d and %d genomes (%.3f s, %.3f s) and extrapolated linearly to %d"
                  % (n_s, bi, t_sk, n1, n2, t1, t2, N),
        "parity": {"sketch_bit_exact": parity_sketch, "counts_bit_exact": parity_counts},
    }


if __name__ == "__main__":
    main()
