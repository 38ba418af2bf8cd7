#!/usr/bin/env python3
"""bench.py -- query throughput of the MI355X NIQKI engine on BASELINE.json's
metric: query genomes/sec + achieved HBM GB/s on a 100k-genome index,
K=31 S=15 W=12 (H=4, J=0.1 -> min_score 3276), synthetic 5 Mbp genomes.

    python bench.py [--gpus N --steps K --warmup W]

One "step" = one pass of the hot path (k-mer rolling hash -> HyperMinHash
sketch -> densification -> gather-histogram over the inverted index ->
threshold + ordered hits) over one batch of query genomes whose bases are
already resident in HBM (a ring of distinct batches, step i uses batch i mod
ring).  On one GPU the step is the C ABI's niqki_query_ahead + niqki_sketch_ahead:
batch i + 1's sketch kernel runs on the handle's sketch lane beside batch i's
gather and hit kernels (K sketch launches and K queries inside the K timed steps;
--no-overlap: one after the other).  N > 1 (torch.distributed.run, one rank per GPU): the index is sharded by
sketch-slot range, the exchange (RCCL all-to-all of sketch slices, sparse
candidate exchange or reduce-scatter of the packed hit vectors) runs inside
libniqki_hip.so (niqki_group_*); torch.distributed only carries the group id
and the timing barrier.  Every rank brings its own --batch query genomes per step ("scaling": "weak"; --scaling
strong cuts one batch over the ranks).

Prints ONE JSON line (rank 0).  `roofline` is for the gather-histogram kernel
(the HBM-bound kernel SURVEY.md 8d grades), timed live with HIP events on the
engine's stream in a few steps of the same run WITHOUT the overlapped sketch kernel
(beside it the launch time is no roofline figure); `kernels` lists every kernel class; `sketch_kernel` holds the
ALU-bound sketch kernel against integer-ALU ceilings measured in this run;
`cpu_baseline` is the oracle (a port of the reference's CPU path) on this
host's cores; `extra_workloads` (1 GPU) are BASELINE.json configs[1], configs[4]
and the matrix path, each with its own in-run parity check.

    python bench.py --shard-of 8      one GPU plays rank 0 of an 8-GPU slot shard
                                      (slots [0, F/8), the full query batch): the
                                      compute half of the 1 -> 8 scaling curve

The whole default run keeps to a wall-clock budget (--budget-s, default 140 s from the start of the process): the
headline, `roofline` and `cpu_baseline` always run, every other leg starts only while its time is left, child processes
get min(their own limit, what is left), and `budget.dropped` names what did not run.
"""
import argparse
import json
import os
import sys
import time

T_START = time.time()

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bench_support import (BENCH_K, HBM_PEAK_GBS, Budget, genome_spec, host_cpu_info, launch_ranks, log, measure_counters,  # noqa: E402,F401
                           query_spec, roofline_record)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=9)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genomes", type=int, default=100_000)
    ap.add_argument("--batch", type=int, default=4096, help="query genomes per step: per GPU (N = 1, and N > 1 with --scaling weak), "
                                                            "or of the whole job (--scaling strong, --shard-of)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = every rank brings --batch query genomes per step (default), strong = one batch of "
                         "--batch queries is cut over the ranks")
    ap.add_argument("--ring", type=int, default=3,
                    help="distinct query batches kept resident in HBM; step i uses batch i mod ring")
    ap.add_argument("--len", type=int, default=5_000_000)
    ap.add_argument("--family", type=int, default=100)
    ap.add_argument("--seed", type=int, default=20261003)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra workloads (configs[1], configs[4], matrix)")
    ap.add_argument("--exchange", default=os.environ.get("NIQKI_EXCHANGE", "auto"),
                    help="cross-shard sum: auto | sparse | reduce_scatter (include/niqki_hip.h, niqki_group_set_option)")
    ap.add_argument("--shard-of", type=int, default=0,
                    help="one GPU as rank 0 of a slot shard of this many GPUs (no exchange; see the module docstring)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not measure the gather path's HBM traffic with rocprofv3 --pmc passes before the run "
                         "(roofline.traffic then comes from profiles/gather_traffic.json)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-legs", action="store_true",
                    help="only the index build, the warm-up and the K timed steps: no end_to_end_d2h / pipelined_step legs, no "
                         "ALU probes, no CPU baseline, no extra workloads (the command tools/profile_round.sh traces, so that a "
                         "kernel's figures in profiles/ are those of the timed launches)")
    ap.add_argument("--transport", default=os.environ.get("NIQKI_GROUP_TRANSPORT", "auto"),
                    help="exchange transport for N > 1: auto (rccl; ipc when the ranks share devices) | rccl | ipc "
                         "(direct peer access through HIP IPC handles, include/niqki_hip.h)")
    ap.add_argument("--devices", type=int, default=0,
                    help="N > 1: deal the ranks over the first D devices only (0 = all visible ones); with fewer devices than "
                         "ranks the ranks share them over the ipc transport (config.ranks_share_devices)")
    ap.add_argument("--pipeline", action="store_true", help=argparse.SUPPRESS)      # (the default since round 6)
    ap.add_argument("--priority-streams", action="store_true",
                    help="N = 1: the handle's stream at the top of the device's stream priority range, its sketch lane at the "
                         "bottom (option stream_priority of the C ABI)")
    ap.add_argument("--gather-cus", type=int, default=0, help=argparse.SUPPRESS)     # experiment: the device split in two
    ap.add_argument("--no-overlap", action="store_true",
                    help="do not run the next batch's sketch kernel beside the gather and hit kernels (N = 1) / the exchange "
                         "(N > 1) of the current one")
    ap.add_argument("--budget-s", type=float, default=float(os.environ.get("NIQKI_BENCH_BUDGET_S", "140")),
                    help="wall-clock budget of the whole run in seconds, from the start of the process (see the module docstring)")
    ap.add_argument("--verify", action="store_true", help=argparse.SUPPRESS)        # (the default for N > 1 since round 6)
    ap.add_argument("--no-verify", action="store_true",
                    help="N > 1: do NOT check every rank's hit lists of its last step against a whole-range handle that it builds "
                         "beside its shard (all genomes sketched once more on every rank: seconds at 100 000 genomes)")
    args = ap.parse_args()
    if args.pmc_child:
        args.no_legs = args.no_overlap = True      # (counters per kernel: every launch has the device to itself)
    if args.no_legs:
        args.no_cpu = args.no_extra = args.no_pmc = True
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: this process becomes one (it has not touched the GPU, torch is not imported)
        sys.exit(launch_ranks(args.gpus))

    # HBM bytes per launch of the gather path, measured now: two rocprofv3 --pmc passes (FETCH_SIZE,
    # WRITE_SIZE: passes of their own, no trace flags) over a 3-launch run of this same command, as
    # child processes started BEFORE this process touches the GPU.  Full default runs on one GPU only.
    budget = Budget(args.budget_s, T_START)
    live_traffic = None
    if (args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.shard_of and not args.no_cpu
            and not args.no_extra and not args.no_pmc and os.environ.get("NIQKI_BENCH_PMC", "1") != "0"
            and budget.want("pmc_passes", 90)):
        live_traffic = measure_counters(args, budget)
        budget.lap("pmc_passes")

    # Only the JSON line may reach stdout: libraries (RCCL prints a version banner)
    # get stderr for the whole run, the result is written to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import niqki_amd
    from niqki_amd.dist import ShardedQuery

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("warning: --gpus %d but WORLD_SIZE %d: the launcher's world size counts" % (args.gpus, world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (before the first HIP call of this process)
    # ranks that share a device (more ranks than GPUs: a one-GPU box rehearsing N > 1): RCCL refuses that,
    # the library's ipc transport does not; torch.distributed then runs over gloo
    n_dev = max(1, torch.cuda.device_count())
    if args.devices > 0:
        n_dev = min(n_dev, args.devices)
    shared = world > n_dev
    transport = args.transport if args.transport != "auto" else ("ipc" if shared else "rccl")
    # (rehearsal of the fall-back below on a one-GPU box: the rccl group "fails" before RCCL is touched)
    inject = os.environ.get("NIQKI_BENCH_INJECT_RCCL_FAILURE") == "1" and args.transport == "auto"
    if inject:
        transport = "rccl"
    if transport == "ipc":
        os.environ["NIQKI_GROUP_TRANSPORT"] = "ipc"
    else:
        os.environ.pop("NIQKI_GROUP_TRANSPORT", None)
    local_dev = local_rank % n_dev
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    # NIQKI_FORCE_DIST=1 runs the sharded (collective) code path even on one rank
    use_dist = world > 1 or os.environ.get("NIQKI_FORCE_DIST") == "1"
    emu = args.shard_of if (args.shard_of > 1 and not use_dist) else 0
    on_gloo = use_dist and shared
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if on_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    K, S, W, H, J = 31, 15, 12, 4, 0.1
    F = 1 << S
    N, L = args.genomes, args.len
    n_fam = max(1, N // args.family)
    G = emu if emu else world                  # shards the index is cut into
    sb, se = niqki_amd.group_slot_range(0 if emu else rank, G, S)
    eng = niqki_amd.Engine(K=K, S=S, W=W, H=H, J=J, device=local_dev, slot_begin=sb, slot_end=se)
    if args.priority_streams and not use_dist and not emu:
        eng.set_option("stream_priority", 1)
        torch.cuda.set_stream(torch.cuda.ExternalStream(eng.get_stream(), device=dev))   # torch's current stream IS the handle's
    else:
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_option("record_len_hint", L)
    stride_b = L  # records are stored back to back; NIQKI_SEQ_PAD bytes follow the last one

    def dev_u32(a):
        return torch.from_numpy(a.astype(np.int64)).to(torch.int32).to(dev)

    def rec_offsets(n):
        return torch.from_numpy((np.arange(n + 1, dtype=np.int64) * stride_b)).to(dev)

    # The exchange transport is chosen once for all ranks.  "auto" on distinct devices means RCCL; should its
    # group not come up, or its first exchange (round 0 of the index build, before anything is inserted) raise,
    # on ANY rank, all ranks agree (one all_reduce over torch.distributed) to go on with the library's ipc
    # transport -- peers' exchange buffers mapped over xGMI -- instead of losing the run.
    def make_group(tr):
        if tr == "rccl" and inject:
            raise RuntimeError("injected failure (NIQKI_BENCH_INJECT_RCCL_FAILURE)")
        if tr == "ipc":
            os.environ["NIQKI_GROUP_TRANSPORT"] = "ipc"
        else:
            os.environ.pop("NIQKI_GROUP_TRANSPORT", None)
        return ShardedQuery(eng, N, F, dev, exchange=args.exchange, cand_cap=256)

    def all_ok(ok):
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if on_gloo else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    can_fall_back = use_dist and world > 1 and args.transport == "auto" and transport == "rccl"
    transport_note = None

    def guarded(what, fn):
        """fn() on every rank; False when it raised somewhere and the ranks agreed to fall back"""
        nonlocal sq, transport, transport_note
        err = None
        try:
            fn()
        except Exception as e:      # noqa: BLE001 -- whatever the transport raised: decided collectively below
            err = e
        if all_ok(err is None):
            return True
        if not can_fall_back or transport != "rccl":
            raise err if err is not None else RuntimeError("%s failed on another rank" % what)
        log("[rank %d] %s failed over rccl (%s): all ranks fall back to the ipc transport" % (rank, what, err))
        transport_note = "rccl: %s failed (%s); ipc used instead" % (what, str(err)[:200] if err else "on another rank")
        if sq is not None:
            try:
                sq.close()
            except Exception:       # noqa: BLE001
                pass
        transport = "ipc"
        sq = make_group("ipc")
        return False

    sq = None
    if use_dist:
        def _mk():
            nonlocal sq
            sq = make_group(transport)
        guarded("group creation", _mk)

    # ---- index build (not timed): synth -> sketch -> (slice exchange) -> insert ----
    t0 = time.time()
    eng.reserve(N)
    GB = 256
    seqbuf = torch.zeros(GB * stride_b + niqki_amd.SEQ_PAD, dtype=torch.uint8, device=dev)
    skbuf = torch.full((GB, F), -1, dtype=torch.int32, device=dev)
    ro_full = rec_offsets(GB)
    n_rounds = ((N + GB - 1) // GB + world - 1) // world
    for r in range(n_rounds):
        g0 = (r * world + rank) * GB           # this rank's GB genomes of the round (rank major = id order)
        n = max(0, min(GB, N - g0))
        if n:
            fam, mem, rate = genome_spec(np.arange(g0, g0 + n), n_fam, args.family)
            eng.synth_dev(args.seed, dev_u32(fam), dev_u32(mem), dev_u32(rate), n, L, stride_b, seqbuf)
            eng.sketch_dev(seqbuf, ro_full if n == GB else rec_offsets(n), n, skbuf)
        if use_dist:
            n_round = max(0, min(world * GB, N - r * world * GB))
            if r == 0 and can_fall_back:
                def _first():
                    sq.insert(skbuf, n_round)
                    eng.synchronize()
                if not guarded("the first slice exchange", _first):
                    sq.insert(skbuf, n_round)      # nothing was inserted: again over ipc
            else:
                sq.insert(skbuf, n_round)
        elif n:
            eng.insert_dev(skbuf, n)
    eng.build()
    eng.synchronize()
    t_index = time.time() - t0
    log("[rank %d] index: %d genomes, slots [%d, %d), tile %d, built in %.1f s" % (rank, eng.n_genomes, sb, se, eng.tile_genomes(), t_index))
    del seqbuf

    # ---- query inputs resident in HBM: a ring of distinct batches, this rank's share ----
    # N > 1: by default every rank brings --batch queries of its own (weak scaling: the step's fixed costs -- the
    # exchange's launches and collective calls -- stay what they are while every shard's gather sees world x batch
    # query slices); --scaling strong cuts ONE batch of --batch queries over the ranks instead
    weak = use_dist and world > 1 and args.scaling == "weak"
    per = args.batch if weak else (args.batch + G - 1) // G            # queries this GPU sketches per step
    nq_all = per * G                           # queries its gather kernel sees per step
    nq_gather = nq_all if (use_dist or emu) else per
    n_steps_all = args.warmup + args.steps
    n_batches = max(1, min(n_steps_all, args.ring))
    qseq = torch.zeros(n_batches * per * stride_b + niqki_amd.SEQ_PAD, dtype=torch.uint8, device=dev)
    for bi in range(n_batches):
        q = bi * nq_all + (0 if emu else rank) * per + np.arange(per)
        fam, mem, rate = query_spec(q, n_fam)
        eng.synth_dev(args.seed, dev_u32(fam), dev_u32(mem), dev_u32(rate), per, L, stride_b,
                      qseq[bi * per * stride_b:])
    d_ro = rec_offsets(per)
    qsk = torch.empty((n_batches, per, F), dtype=torch.int32, device=dev)
    cap = per * 4096
    hit_off = torch.zeros((n_steps_all, per + 1), dtype=torch.int64, device=dev)
    hc = torch.zeros(cap, dtype=torch.int32, device=dev)
    hg = torch.zeros(cap, dtype=torch.int32, device=dev)
    stride = niqki_amd.row_stride(N)
    allsk = None
    if emu:
        # the other ranks' sketches of every batch, made once outside the timed region: in the real
        # job they arrive through the slice exchange
        allsk = torch.empty((n_batches, nq_all, F), dtype=torch.int32, device=dev)
        tmp = torch.zeros(per * stride_b + niqki_amd.SEQ_PAD, dtype=torch.uint8, device=dev)
        for bi in range(n_batches):
            for r in range(G):
                fam, mem, rate = query_spec(bi * nq_all + r * per + np.arange(per), n_fam)
                eng.synth_dev(args.seed, dev_u32(fam), dev_u32(mem), dev_u32(rate), per, L, stride_b, tmp)
                eng.sketch_dev(tmp, d_ro, per, allsk[bi, r * per:(r + 1) * per])
        del tmp
        EC, ESC = 256, 1024                       # candidate / survivor capacities (the group's defaults)
        cand = torch.zeros((nq_all, EC), dtype=torch.int32, device=dev)
        ncand = torch.zeros(nq_all, dtype=torch.int32, device=dev)
        surv = torch.zeros((nq_all, ESC, 2), dtype=torch.int32, device=dev)
        nsurv = torch.zeros(nq_all, dtype=torch.int32, device=dev)
        # stand-in for the all-gathered candidate lists of the G ranks: this rank's own list G times
        cand_all = torch.zeros((nq_all, G, EC), dtype=torch.int32, device=dev)
        mine = torch.zeros((nq_all, G * EC), dtype=torch.int16, device=dev)
        e_thr = -(-eng.min_score // G)
    eng.synchronize()

    # N = 1: batch i + 1's sketch kernel runs on the handle's sketch lane beside batch i's gather and hit kernels
    # (niqki_sketch_ahead / niqki_query_ahead: one handle, the overlap inside the C ABI).
    # N > 1: batch i's exchange (slices, candidate lists, sums: the GPU mostly waits for its peers) runs
    # beside batch i+1's sketch kernel: a second handle sketches on a side stream, niqki_group_query_begin
    # returns without waiting, niqki_group_query_end is the step's one host wait.
    ahead = not use_dist and not emu and not args.no_overlap
    if args.gather_cus and not use_dist and not emu:
        # experiment (profiles/r06_cu_split.txt): the handle's stream limited to the low --gather-cus compute units of
        # the device's numbering (one in eight per XCD), the sketch lane to the others
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        words = (ctypes.c_uint32 * ((cus + 31) // 32))()
        for c in range(min(args.gather_cus, cus)):
            words[c // 32] |= 1 << (c % 32)
        masked = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(masked), len(words), words)
        assert rc == 0 and masked.value, rc
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.ExternalStream(masked.value, device=dev))
        eng.set_stream(masked.value)
        if ahead and args.gather_cus < cus:
            eng.set_option("sketch_lane_cus", cus - args.gather_cus)
    overlap = use_dist and not args.no_overlap
    if overlap:
        sk_eng = niqki_amd.Engine(K=K, S=S, W=W, H=H, J=J, device=local_dev)
        side = torch.cuda.Stream(device=dev)
        sk_eng.set_stream(side.cuda_stream)
        sk_eng.set_option("record_len_hint", L)
        ev_sk = [torch.cuda.Event() for _ in range(n_batches)]      # sketches of ring slot b are complete
        ev_used = [torch.cuda.Event() for _ in range(n_batches)]    # ... have been consumed by their batch
        sketched = [-1] * n_batches                                   # step whose sketches slot b holds

        def sketch_ahead(sj):
            bj = sj % n_batches
            # (the last timed step sketches one batch beyond the run, as the first one found its own
            # batch sketched by the warm-up: K sketch launches inside the K timed steps)
            if sj > n_steps_all or sketched[bj] == sj:
                return
            with torch.cuda.stream(side):
                if sketched[bj] >= 0:
                    side.wait_event(ev_used[bj])
                sk_eng.sketch_dev(qseq[bj * per * stride_b:], d_ro, per, qsk[bj])
                ev_sk[bj].record(side)
            sketched[bj] = sj

    def serial_step(si, off, c_, g_):
        """sketch, then query, on the handle's stream (what --no-overlap times, and the roofline steps)"""
        bi = si % n_batches
        eng.sketch_dev(qseq[bi * per * stride_b:], d_ro, per, qsk[bi])
        eng.query_dev(qsk[bi], per, off, c_, g_, cap)

    if ahead:
        eng.sketch_ahead_dev(qseq, d_ro, per)          # batch 0, before the warm-up (or the first timed step)

    def step(si):
        bi = si % n_batches
        if ahead:
            # the hits of batch si (its sketches were made beside the step before), then batch si + 1's sketch kernel
            # onto the sketch lane: K queries and K sketch launches inside K steps
            eng.query_ahead_dev(hit_off[si], hc, hg, cap)
            eng.sketch_ahead_dev(qseq[((si + 1) % n_batches) * per * stride_b:], d_ro, per)
            return
        if overlap:
            sketch_ahead(si)                       # (only the first step finds its batch not sketched yet)
            torch.cuda.current_stream().wait_event(ev_sk[bi])
            sq.begin(qsk[bi], hit_off[si], hc, hg, cap)
            ev_used[bi].record(torch.cuda.current_stream())
            sketch_ahead(si + 1)
            sq.end()
            return
        if use_dist:
            eng.sketch_dev(qseq[bi * per * stride_b:], d_ro, per, qsk[bi])
            sq.step(qsk[bi], hit_off[si], hc, hg, cap)
        elif emu:
            eng.sketch_dev(qseq[bi * per * stride_b:], d_ro, per, qsk[bi])
            # rank 0's compute of one step of the sparse exchange: its share sketched above; the gather of ALL queries
            # over its slots leaves candidates and survivors, no counter rows; the (stand-in) candidate lists of all
            # ranks are looked up in the survivors; the hits of its own queries come from the candidates' counts
            eng.query_survivors_dev(allsk[bi], nq_all, e_thr, max(1, e_thr // 2), EC, ESC, cand, ncand, surv, nsurv)
            cand_all.copy_(cand.unsqueeze(1).expand(nq_all, G, EC))
            eng.survivor_counts_dev(allsk[bi], nq_all, cand_all, G * EC, surv, nsurv, ESC, mine)
            eng.hits_from_candidates_dev(cand_all, mine, per, G * EC, hit_off[si], hc, hg, cap)
        else:
            serial_step(si, hit_off[si], hc, hg)

    def barrier():
        if use_dist:
            dist.barrier()

    for bi in range(args.warmup):
        step(bi)
    eng.synchronize()
    torch.cuda.synchronize()
    eng.profile(True)
    eng.profile_reset()
    if overlap:
        sk_eng.profile(True)
        sk_eng.profile_reset()
    barrier()
    torch.cuda.synchronize()
    # per-step device times beside the wall clock of the K steps: events on torch's current stream (the
    # engine's stream) at the step boundaries, read after the run
    ev_step = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev_step[0].record()
    for k, si in enumerate(range(args.warmup, n_steps_all)):
        step(si)
        ev_step[k + 1].record()
    eng.synchronize()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    step_ms = sorted(ev_step[k].elapsed_time(ev_step[k + 1]) for k in range(args.steps))
    ms_median = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if on_gloo else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    prof = {name: eng.profile_read(kc) for name, kc in (
        ("sketch", niqki_amd.KC_SKETCH), ("densify", niqki_amd.KC_DENSIFY), ("gather", niqki_amd.KC_GATHER),
        ("hits", niqki_amd.KC_HITS), ("exchange", niqki_amd.KC_EXCHANGE))}
    eng.profile(False)
    if overlap:
        prof["sketch"] = sk_eng.profile_read(niqki_amd.KC_SKETCH)
        sk_eng.profile(False)

    # N > 1 with the next batch sketched beside the exchange: the gather launches of the timed steps share the
    # device with a sketch kernel, their durations say nothing about the kernel.  Its roofline figure comes from a
    # few more steps of this same run without the overlapped sketch (every rank takes them: the exchange is collective).
    roofline_from = None
    kernels_overlapped = None
    roof_steps = list(range(args.warmup, n_steps_all))     # the steps whose gather launches `roofline` is taken from
    roof_gather = None
    if ahead:
        # one GPU: the gather launches of the timed steps ran beside the next batch's sketch kernel.  The kernels'
        # own times -- `roofline`, `kernels`, `sketch_kernel` -- come from n_r more steps of this run that sketch and
        # query one after the other on the handle's stream (hits into buffers of their own: hc / hg keep the last
        # timed step's for the parity legs); these also leave every ring batch's sketches in qsk
        kernels_overlapped = {k: {"ms": round(v[0], 3), "launches": v[1]} for k, v in prof.items()}
        off2 = torch.zeros(per + 1, dtype=torch.int64, device=dev)
        hc2, hg2 = torch.zeros_like(hc), torch.zeros_like(hg)
        eng.query_ahead_dev(off2, hc2, hg2, cap)        # (the batch the last timed step sketched beyond the run)
        n_r = max(min(5, args.steps), n_batches)
        serial_step(args.warmup, off2, hc2, hg2)        # (warm-up of this form: its first launch allocates)
        eng.synchronize()
        eng.profile(True)
        eng.profile_reset()
        ts = time.perf_counter()
        for si in range(args.warmup, args.warmup + n_r):
            serial_step(si, off2, hc2, hg2)
        eng.synchronize()
        t_serial = (time.perf_counter() - ts) / n_r
        sprof = {name: eng.profile_read(kc) for name, kc in (
            ("sketch", niqki_amd.KC_SKETCH), ("densify", niqki_amd.KC_DENSIFY), ("gather", niqki_amd.KC_GATHER),
            ("hits", niqki_amd.KC_HITS), ("exchange", niqki_amd.KC_EXCHANGE))}
        eng.profile(False)
        # scaled to the K timed steps, like the figures they stand in for
        prof = {k: (v[0] * args.steps / n_r, v[1] * args.steps // n_r) for k, v in sprof.items()}
        roof_steps, roof_gather = list(range(args.warmup, args.warmup + n_r)), sprof["gather"]
        roofline_from = ("%d more steps of this run with the sketch kernel BEFORE the query instead of beside it (%.2f ms per step); "
                         "the timed steps' launches share the device with the next batch's sketch kernel" % (n_r, t_serial * 1e3))
        del off2, hc2, hg2
    if use_dist and overlap:
        n_r = min(3, args.steps)
        eng.profile(True)
        eng.profile_reset()
        for si in range(args.warmup, args.warmup + n_r):
            bi = si % n_batches
            eng.sketch_dev(qseq[bi * per * stride_b:], d_ro, per, qsk[bi])
            sq.step(qsk[bi], hit_off[si], hc, hg, cap)
        eng.synchronize()
        g_ms, g_n = eng.profile_read(niqki_amd.KC_GATHER)
        eng.profile(False)
        barrier()
        if g_n:
            prof["gather"] = (g_ms * args.steps / n_r, g_n * args.steps // n_r)
            roofline_from = "%d more steps of this run without the overlapped sketch kernel (the timed steps' gather launches share the device with it)" % n_r

    # --verify (N > 1, small indexes): this rank's hit lists of the last timed step against a whole-range
    # handle that holds every genome of the index
    verify = None
    if use_dist and (args.verify or (world > 1 and not args.no_verify)):
        # The local half may fail on one rank only (an allocation, say): it is caught and counted as "not equal", so
        # that every rank still reaches the two collectives below -- a check beside the line must not hang the job.
        si = n_steps_all - 1
        same, nh, v_err = False, 0, None
        ref = None
        try:
            ref = niqki_amd.Engine(K=K, S=S, W=W, H=H, J=J, device=local_dev)
            ref.set_stream(torch.cuda.current_stream().cuda_stream)
            ref.set_option("record_len_hint", L)
            vseq = torch.zeros(GB * stride_b + niqki_amd.SEQ_PAD, dtype=torch.uint8, device=dev)
            vsk = torch.full((GB, F), -1, dtype=torch.int32, device=dev)
            for g0 in range(0, N, GB):
                n = min(GB, N - g0)
                fam, mem, rate = genome_spec(np.arange(g0, g0 + n), n_fam, args.family)
                ref.synth_dev(args.seed, dev_u32(fam), dev_u32(mem), dev_u32(rate), n, L, stride_b, vseq)
                ref.sketch_dev(vseq, rec_offsets(n), n, vsk)
                ref.insert_dev(vsk, n)
            del vseq, vsk
        except Exception as e:      # noqa: BLE001
            v_err = "whole-range handle: %s" % str(e)[:200]
        # (hc / hg hold whatever step ran last -- the extra roofline steps above may have: the last timed step again,
        # through the group, on every rank)
        sq.step(qsk[si % n_batches], hit_off[si], hc, hg, cap)
        eng.synchronize()
        if v_err is None:
            try:
                r_off = torch.zeros(per + 1, dtype=torch.int64, device=dev)
                r_hc, r_hg = torch.zeros(cap, dtype=torch.int32, device=dev), torch.zeros(cap, dtype=torch.int32, device=dev)
                ref.query_dev(qsk[si % n_batches], per, r_off, r_hc, r_hg, cap)
                ref.synchronize()
                nh = int(r_off[per].item())
                same = bool(torch.equal(r_off, hit_off[si]) and torch.equal(r_hc[:nh], hc[:nh]) and torch.equal(r_hg[:nh], hg[:nh]))
                del r_off, r_hc, r_hg
            except Exception as e:      # noqa: BLE001
                v_err = "whole-range query: %s" % str(e)[:200]
        if v_err:
            log("[rank %d] --verify: %s" % (rank, v_err))
        ok = torch.tensor([1 if same else 0, 0 if v_err is None else 1], dtype=torch.int32, device="cpu" if on_gloo else dev)
        dist.all_reduce(ok, op=dist.ReduceOp.SUM)
        verify = {"hit_lists_equal_whole_range_handle": int(ok[0].item()) == world, "ranks_equal": int(ok[0].item()),
                  "ranks_that_could_not_check": int(ok[1].item()), "queries_per_rank": per, "hits_rank0": nh,
                  "rank0_note": v_err}
        if ref is not None:
            ref.close()

    # ---- roofline of the gather kernel: algorithmic bytes 4T + 20F per query ----
    f_local = se - sb
    T = 0
    for si in roof_steps:
        bi = si % n_batches
        if use_dist:   # measurement only: whole sketches of every rank, this shard's slots are what counts
            if on_gloo:
                fc = torch.empty((world * per, F), dtype=torch.int32)
                dist.all_gather_into_tensor(fc.view(-1), qsk[bi].reshape(-1).cpu())
                full = fc.to(dev)
            else:
                full = torch.empty((world * per, F), dtype=torch.int32, device=dev)
                dist.all_gather_into_tensor(full.view(-1), qsk[bi].reshape(-1))
            T += int(eng.gathered_dev(full, world * per).sum())
        else:
            T += int(eng.gathered_dev(allsk[bi] if emu else qsk[bi], nq_gather).sum())
    n_q_local = len(roof_steps) * nq_gather
    gather_ms, gather_launches = roof_gather if roof_gather is not None else prof["gather"]
    alg_bytes = 4 * T + 20 * f_local * n_q_local
    # the same quantities at the sizes the layout really stores: 2-byte ids, one 8-byte entry per slot and
    # tile, 2-byte counters written back
    n_tiles = -(-N // max(eng.tile_genomes(), 1))
    layout_min = 2 * T + (8 * n_tiles + 4) * f_local * n_q_local + 2 * N * n_q_local
    achieved = alg_bytes / (gather_ms * 1e-3) / 1e9 if gather_ms > 0 else 0.0
    total_hits = int(hit_off[args.warmup:, per].sum().item())
    overflow = bool((hit_off[:, per] > cap).any().item())

    # HBM bytes of the gather kernel from the committed PMC passes (separate rocprofv3
    # --pmc runs of this same command), unless this run measured it itself (measure_traffic)
    traffic, traffic_source = None, None
    if live_traffic is not None:
        traffic, traffic_source = live_traffic["bytes_per_launch"], live_traffic["source"]
    else:
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "gather_traffic.json")))
            if (tj["index_genomes"], tj["query_batch"], tj["tile_genomes"]) == (N, nq_gather, eng.tile_genomes()) and world == 1 and not emu:
                traffic = tj["traffic_bytes_per_launch"]
                traffic_source = "profiles/gather_traffic.json (committed rocprofv3 --pmc passes of this command)"
        except (OSError, KeyError, ValueError):
            pass

    # ---- what a streaming copy reaches on this device, now (the practical HBM ceiling beside the 8 TB/s spec) ----
    copy_gbs = eng.measure_alu(4) / 1e9 if (rank == 0 and not args.no_legs) else None

    # ---- sketch kernel against integer-ALU ceilings measured now, on this device ----
    kmers = args.steps * per * max(L - K, 0)
    sk_rate = kmers / (prof["sketch"][0] * 1e-3) if prof["sketch"][0] else 0.0
    alu = None
    if rank == 0 and not args.no_legs:
        adds, muls, arith, vop3 = eng.measure_alu(0), eng.measure_alu(1), eng.measure_alu(2), eng.measure_alu(3)
        # vector instructions per k-mer: counted in this run (SQ_INSTS_VALU pass of measure_counters) or, where the
        # run took no counter passes, the committed figure of the same kernel
        if live_traffic is not None and live_traffic.get("valu"):
            valu_per_kmer = live_traffic["valu"]["valu_per_kmer"]
            valu_source = ("SQ_INSTS_VALU x 64 over the k-mers of the %d sketch launches of a rocprofv3 --pmc pass of this "
                           "same command, run by bench.py before its timed run" % live_traffic["valu"]["launches"])
        else:
            valu_per_kmer = 27.8
            valu_source = "profiles/r03_sketch_sq_counters.txt (not measured in this run)"
        alu = {"add_lane_ops_per_s": adds, "mul_lane_ops_per_s": muls, "vop3_lane_ops_per_s": vop3,
               "arithmetic_only_kmers_per_s": arith,
               "alu_frac": sk_rate / arith if arith else None,
               "valu_per_kmer": valu_per_kmer, "valu_per_kmer_source": valu_source,
               "issue_frac": sk_rate * valu_per_kmer / vop3 if vop3 else None,
               "note": "alu_frac = sketch kernel k-mers/s over the rate of its per-k-mer arithmetic alone (roll, canonical "
                       "choice, filter hash; no LDS table, compaction, candidates or memory); issue_frac = its vector "
                       "instructions per second (k-mers/s x valu_per_kmer) over the measured issue rate of v_lshl_add_u32, "
                       "which every vector opcode but add/and/or/xor/mov shares on gfx950 (profiles/r03_opcode_costs.txt); "
                       "the four ALU rates are measured in this run, valu_per_kmer as valu_per_kmer_source says"}

    # bytes a rank sends per step in the exchange (sketch slices, then the candidate lists or the
    # dense counters): with the step time this bounds the average xGMI rate per GPU
    xbytes = None
    if use_dist:
        g1 = (world - 1) / world
        xbytes = nq_all * (F // world) * 2 * g1                          # all-to-all of F/G-slot sketch slices (int16)
        if sq.exchange == "sparse":
            xbytes += nq_all * sq.cand_cap * 4 * g1 + nq_all * 4 * g1    # all-gather of candidates + their counts
            xbytes += nq_all * world * sq.cand_cap * 2 * g1              # reduce-scatter of the candidates' partial counts (u16)
        else:
            xbytes += nq_all * (stride // 2) * 4 * g1                    # dense u16 counters as u32 pairs
    if rank == 0:
        n_queries = args.steps * nq_all
        out = {
            "metric": "query genomes/sec, %dk-genome index, K=31 S=15 W=12" % (N // 1000),
            "value": n_queries / dt,
            "unit": "genomes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_median": ms_median,
            "value_at_median_step": nq_all / (ms_median * 1e-3),
            "step_ms_min_max": [step_ms[0], step_ms[-1]],
            "higher_is_better": True,
            "scaling": "weak" if (weak or world == 1) else "strong",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": "%d synthetic %d bp genomes indexed (families of %d, 0.1-5%% substitutions), "
                            "%d query genomes per step resident in HBM, K=31 S=15 W=12 H=4 J=0.1"
                            % (N, L, args.family, nq_all),
                "index_genomes": N, "query_batch": nq_all, "genome_len": L,
                "parallelism": ("slot-shard x%d (%s exchange, %s transport inside libniqki_hip.so%s)"
                                % (world, sq.exchange, sq.transport, ", next batch sketched beside the exchange" if overlap else "")) if use_dist
                else ("1 GPU as rank 0 of a %d-way slot shard" % emu if emu else
                      ("1 GPU, batch i + 1's sketch kernel beside batch i's gather and hit kernels (niqki_sketch_ahead / niqki_query_ahead)"
                       if ahead else "1 GPU, sketch kernel then query on one stream")),
                "sketch_beside_query": bool(ahead or overlap),
                "transport": sq.transport if use_dist else None,
                "transport_note": transport_note,
                "ranks_share_devices": bool(shared) if use_dist else None, "devices_visible": n_dev,
                "exchange_redone_densely": sq.overflows if use_dist else 0,
                "exchange_bytes_per_rank_per_step": xbytes,
                "exchange_avg_gbs_per_rank": (xbytes / (dt / args.steps) / 1e9) if xbytes else None,
                # did the transport itself see N ranks?  (the communicator's size as RCCL reports it / the peers whose
                # sequence words this rank has mapped over ipc: niqki_group_get_stat "ranks_seen")
                "transport_ranks_seen": sq.ranks_seen if use_dist else None,
                # bytes a rank sends to ONE peer per step over the device time of the step's exchange-class spans on rank 0
                # (slice packing, the collectives, candidate look-ups: NIQKI_KC_EXCHANGE) -- xGMI is point to point, one
                # link per peer (153 GB/s each); a lower bound of the link rate while data moves
                "exchange_ms_per_step": (prof["exchange"][0] / max(1, args.steps)) if use_dist else None,
                "exchange_gbs_per_link": (xbytes / max(1, world - 1) / (prof["exchange"][0] / max(1, args.steps) * 1e-3) / 1e9)
                if (xbytes and use_dist and world > 1 and prof["exchange"][0] > 0) else None,
                "tile_genomes": eng.tile_genomes(), "index_build_s": round(t_index, 2),
                "hits_per_query": total_hits / max(1, args.steps * per), "hit_overflow": overflow,
            },
            "roofline": roofline_record(achieved, traffic, traffic_source, live_traffic, gather_ms, gather_launches, alg_bytes, layout_min,
                                        copy_gbs, roofline_from, T, n_q_local, False),
            "kernels": {k: {"ms": round(v[0], 3), "launches": v[1]} for k, v in prof.items()},
            # the sketch kernel is integer-ALU bound (4 64-bit multiplies per k-mer):
            # its rate in k-mers, the HBM bytes it needs (1 byte per base + the sketch) and the ALU ceilings
            "sketch_kernel": {
                "gkmers_per_s": sk_rate / 1e9,
                "hbm_gbs": (args.steps * per * (L + 4 * F)) / (prof["sketch"][0] * 1e-3) / 1e9 if prof["sketch"][0] else 0.0,
                "bound": "valu", "alu": alu,
            },
            "cpu_baseline": None,
            "end_to_end_d2h": None,
        }
        if ahead:
            out["kernels_note"] = ("kernels / roofline / sketch_kernel: each kernel's own time, from %d steps of this run that sketch and query "
                                   "one after the other (scaled to %d steps); kernels_beside_each_other: the same classes inside the timed "
                                   "steps, where batch i + 1's sketch kernel shares the device with batch i's gather and hit kernels"
                                   % (len(roof_steps), args.steps))
            out["kernels_beside_each_other"] = kernels_overlapped
            out["serial_step"] = {"ms_per_step": t_serial * 1e3, "value": per / t_serial, "unit": "genomes/s", "steps": len(roof_steps),
                                  "note": "sketch kernel, then the query, on one stream (bench.py --no-overlap times this as the line)"}

        if emu:
            # what one shard of the real job computes per step; the exchange (nq * F/G * 2 bytes of slices out,
            # candidate lists) is not part of it
            out["metric"] += " (one GPU as rank 0 of %d slot shards, compute only)" % emu
            out["shard_emulation"] = {
                "shards": emu, "slots": [sb, se], "queries_sketched_per_step": per, "queries_gathered_per_step": nq_all,
                "gather_ms_per_step": gather_ms / max(1, gather_launches), "roofline_frac_algorithmic": achieved / HBM_PEAK_GBS,
                "roofline_frac_layout_min": layout_min / max(1, gather_launches) / (gather_ms / max(1, gather_launches) * 1e-3) / 1e9 / HBM_PEAK_GBS if gather_ms else None,
                "counter_row_bytes_per_query": 0,
                "counter_bytes_written_per_step": int(nsurv.sum().item()) * 8 + int(ncand.sum().item()) * 4 + 8 * nq_all,
                "counter_bytes_written_per_step_with_rows": 2 * N * nq_all,
                "survivors_per_query": float(nsurv.float().mean().item()), "candidates_per_query": float(ncand.float().mean().item()),
                "sketch_ms_per_step": prof["sketch"][0] / max(1, args.steps), "hits_ms_per_step": prof["hits"][0] / max(1, args.steps),
                "projected_genomes_per_s_if_exchange_is_free": nq_all / (dt / args.steps),
            }
        if verify is not None:
            out["verify"] = verify

        # ---- everything below is optional: the line above is complete, the legs fill it in while the budget lasts ----
        emitted = []

        def emit(hard_stop=False):
            if emitted:
                return
            emitted.append(1)
            out["budget"] = budget.record()
            if hard_stop:
                out["budget"]["hard_stop"] = True
            os.write(real_stdout, (json.dumps(out) + "\n").encode())
        budget.lap("index_steps_roofline")
        if world == 1 and not emu and not args.no_legs:
            budget.arm(45.0, emit)
            legs(out, budget, eng, niqki_amd, torch, dev, args, qseq, qsk, hit_off, hc, hg, stride_b, L, N, per, (K, S, W, H, J), cap,
                 n_batches, d_ro, ms_median, ahead)
            budget.disarm()
        emit()
    if use_dist:
        sq.close()
    if overlap:
        sk_eng.close()
    eng.close()
    if use_dist:
        dist.destroy_process_group()


def legs(out, budget, eng, niqki_amd, torch, dev, args, qseq, qsk, hit_off, hc, hg, stride_b, L, N, per, prm, cap, n_batches, d_ro,
         ms_median, ahead):
    """The legs beside the headline, most important first; each fills its key of `out` and none may cost the line: errors
    are recorded, a leg without time left is dropped (budget.dropped)."""
    def guarded(name, fn):
        try:
            fn()
        except Exception as e:      # noqa: BLE001 -- a leg beside the headline: never the run's failure
            budget.dropped.append({"leg": name, "error": str(e)[:300]})
            log("[bench] %s failed: %s" % (name, e))
        budget.lap(name)

    deferred = []     # the real reference over the whole index (31 s of inserts): after the BASELINE configs, if time is left
    if not args.no_cpu:
        def _cpu():
            out["cpu_baseline"] = cpu_baseline(eng, niqki_amd, qseq, qsk, hit_off, hc, hg, args, stride_b, L, N, per, prm, deferred)
        guarded("cpu_baseline", _cpu)

    def _d2h():
        # ---- the same steps with the hits copied back to the host (SURVEY.md 8d "end-to-end incl. D2H of hits") ----
        # as a caller would: two sets of hit buffers, step i+1 runs while a copy stream brings step i's
        # hit_off and then exactly its hits into page-locked memory
        if True:
            hcs, hgs = [torch.zeros_like(hc), torch.zeros_like(hc)], [torch.zeros_like(hg), torch.zeros_like(hg)]
            h_off = [torch.empty(per + 1, dtype=torch.int64).pin_memory() for _ in range(2)]
            h_hc = torch.empty(cap, dtype=torch.int32).pin_memory()
            h_hg = torch.empty(cap, dtype=torch.int32).pin_memory()
            cstream = torch.cuda.Stream(device=dev)
            done = [torch.cuda.Event() for _ in range(2)]
            n_d2h = args.steps                            # the same number of steps as the timed line
            nh_tot = 0
            ev_d = [torch.cuda.Event(enable_timing=True) for _ in range(n_d2h + 1)]

            def fetch(k, si):     # hits of step si (buffer set k) to the host, on the copy stream
                nonlocal nh_tot
                with torch.cuda.stream(cstream):
                    cstream.wait_event(done[k])
                    h_off[k].copy_(d_off[si - args.warmup], non_blocking=True)
                    cstream.synchronize()                 # the sizes first ...
                    nh = int(h_off[k][per])
                    h_hc[:nh].copy_(hcs[k][:nh], non_blocking=True)      # ... then exactly the hits
                    h_hg[:nh].copy_(hgs[k][:nh], non_blocking=True)
                    cstream.synchronize()
                nh_tot += nh
            d_off = torch.zeros((n_d2h, per + 1), dtype=torch.int64, device=dev)     # (hit_off keeps the timed steps')
            if ahead:
                eng.sketch_ahead_dev(qseq[(args.warmup % n_batches) * per * stride_b:], d_ro, per)
            eng.synchronize()
            torch.cuda.synchronize()
            td = time.perf_counter()
            ev_d[0].record()
            for j, si in enumerate(range(args.warmup, args.warmup + n_d2h)):
                k = j % 2
                bi = si % n_batches
                if ahead:
                    eng.query_ahead_dev(d_off[j], hcs[k], hgs[k], cap)
                    eng.sketch_ahead_dev(qseq[((si + 1) % n_batches) * per * stride_b:], d_ro, per)
                else:
                    eng.sketch_dev(qseq[bi * per * stride_b:], d_ro, per, qsk[bi])
                    eng.query_dev(qsk[bi], per, d_off[j], hcs[k], hgs[k], cap)
                done[k].record(torch.cuda.current_stream())
                ev_d[j + 1].record()
                if j:
                    fetch(1 - k, si - 1)                  # while step si runs
            fetch((n_d2h - 1) % 2, args.warmup + n_d2h - 1)
            eng.synchronize()
            torch.cuda.synchronize()
            td = time.perf_counter() - td
            if ahead:       # (the batch sketched beyond the leg)
                eng.query_ahead_dev(d_off[0], hcs[0], hgs[0], cap)
                eng.synchronize()
            d_ms = sorted(ev_d[j].elapsed_time(ev_d[j + 1]) for j in range(n_d2h))
            d_med = d_ms[len(d_ms) // 2] if len(d_ms) % 2 else 0.5 * (d_ms[len(d_ms) // 2 - 1] + d_ms[len(d_ms) // 2])
            out["end_to_end_d2h"] = {"value": n_d2h * per / td, "unit": "genomes/s", "ms_per_step": td / n_d2h * 1e3, "steps": n_d2h,
                   "ms_per_step_median": d_med, "step_ms_min_max": [d_ms[0], d_ms[-1]],
                   "value_at_median_step": per / (d_med * 1e-3),
                   # against the timed line's own median step (the same steps without the copies): what the D2H of the hits costs
                   "median_step_over_the_lines": d_med / ms_median if ms_median else None,
                   "hit_bytes_per_step": 8 * (per + 1) + 8 * nh_tot // n_d2h,
                   "note": "the timed step with hit_off, hit_counts and hit_gids copied into page-locked host memory by a copy "
                           "stream while the next step runs (two sets of hit buffers); ms_per_step includes the last step's "
                           "copy, which nothing overlaps, and the host's waits for the sizes"}
            del hcs, hgs, d_off

    if not use_dist_env() and budget.want("end_to_end_d2h", 5):
        guarded("end_to_end_d2h", _d2h)

    def _reference():
        if deferred and out.get("cpu_baseline") and budget.want("cpu_baseline.reference (the reference's own Index over all genomes)", 75):
            def _run():
                out["cpu_baseline"]["reference"] = deferred[0]()
            guarded("cpu_baseline.reference", _run)
    if args.no_extra:
        _reference()
    else:
        extra = out.setdefault("extra_workloads", {})
        extra_workloads(niqki_amd, torch, dev, args, budget, extra, guarded, _reference, no_cpu=args.no_cpu)


def use_dist_env():
    return os.environ.get("NIQKI_FORCE_DIST") == "1"


def torch_index(like, idx):
    import torch
    return torch.from_numpy(np.asarray(idx, dtype=np.int64)).to(like.device)


def cpu_baseline(eng, niqki_amd, qseq, qsk, hit_off, hc, hg, args, stride_b, L, N, per, prm, deferred):
    """The oracle (port of the reference CPU path) on this host, on a bounded
    sample of the same workload; also the in-run parity check (sketches, dense counters and the
    thresholded, ordered hit lists of the sample against what the timed steps produced)."""
    from oracle import pyoracle as po
    K, S, W, H, J = prm
    p = po.make_params(K, S, W, H, J)
    host = host_cpu_info()
    omp_max = po.lib().nqo_max_threads()
    phys = host["physical_cores"] or omp_max
    # thread counts: the physical cores and the logical CPUs (what the host has), the cgroup quota if there is one
    # (what this process gets), and halvings in between: the scaling table says which of them is the limit
    cand = {omp_max, min(omp_max, phys), max(1, min(omp_max, phys) // 2), max(1, min(omp_max, phys) // 4)}
    if host["cgroup_cpu_quota"]:
        qn = max(1, int(round(host["cgroup_cpu_quota"])))
        cand |= {min(omp_max, qn), min(omp_max, 2 * qn)}
    cand = sorted(cand, reverse=True)
    quota_threads = max(1, min(omp_max, int(round(host["cgroup_cpu_quota"])))) if host["cgroup_cpu_quota"] else min(omp_max, phys)
    n_s = int(min(per, max(8, 4 * quota_threads)))       # four query genomes per CPU this job has
    t_leg = {}
    t_mark = [time.perf_counter()]

    def leg_done(name):
        now = time.perf_counter()
        t_leg[name] = round(now - t_mark[0], 2)
        t_mark[0] = now
    si = args.warmup + args.steps - 1  # the last timed step: its hits are what hc / hg still hold
    bi = si % qsk.shape[0]
    # the sample: n_s queries spread evenly over the whole batch (launch positions of every 1024-query group of the
    # look-up pre-pass and of both ends of the locality order), not its first n_s
    pos = (np.arange(n_s, dtype=np.int64) * per) // n_s + (per // n_s) // 2
    base = bi * per * stride_b
    rec = np.stack([qseq[base + int(q) * stride_b: base + int(q) * stride_b + L].cpu().numpy() for q in pos])
    rec_off = (np.arange(n_s + 1) * L).astype(np.uint64)
    # sketch leg at every thread count (each pass sketches all n_s genomes: a few seconds together)
    # the headline runs on the CPUs this job HAS: the cgroup quota where there is one (more threads than that only
    # time-slice), else the physical cores; the other counts make the scaling tables
    leg_done("sample_to_host")
    t_sk, sk_cpu, sk_table = None, None, {}
    for th in cand:
        t0 = time.perf_counter()
        out = po.sketch_batch(p, rec.reshape(-1), rec_off, threads=th)
        t = time.perf_counter() - t0
        sk_table[th] = n_s / t
        if th == quota_threads:
            t_sk, sk_cpu = t, out
    if sk_cpu is None:
        t0 = time.perf_counter()
        sk_cpu = po.sketch_batch(p, rec.reshape(-1), rec_off, threads=quota_threads)
        t_sk = time.perf_counter() - t0
        sk_table[quota_threads] = n_s / t_sk
    cores = quota_threads
    leg_done("sketch_leg")
    sk_gpu = qsk[bi][torch_index(qsk, pos)].cpu().numpy()
    parity_sketch = bool(np.array_equal(sk_cpu, sk_gpu))
    # gather leg: the oracle's query loop timed on EVERY sub-index of <= 16384 genomes the index is cut into
    # (the seven of them hold all 100 000 genomes: their sum is the whole-index figure, nothing extrapolated);
    # the sub-index' arrays are first touched by the threads that gather from them (nqo_index_spread);
    # parity over ALL columns from the same seven sub-indexes
    n_par = min(n_s, 16)
    par = (np.arange(n_par) * n_s) // n_par          # which of the sample: again evenly spread
    exp_cols = np.zeros((n_par, N), np.uint32)
    cnt = eng.query_counts(sk_gpu[par])
    t_q, n_sub_ix, q_table, q_threads = 0.0, 0, {}, cores
    for b0 in range(0, N, 16384):
        n_sub = min(16384, N - b0)
        sub = eng.get_sketches(b0, n_sub)
        ix = po.Index(p, sub)
        if b0 == 0:
            # the first sub-index picks the thread count of the gather leg (random 4-byte reads: memory bound,
            # its best count need not be the sketch leg's)
            for th in cand:
                ix.spread(th)
                ix.query_batch(sk_cpu, threads=th)
                t0 = time.perf_counter()
                ix.query_batch(sk_cpu, threads=th)
                q_table[th] = n_s / (time.perf_counter() - t0)
            q_threads = quota_threads if quota_threads in q_table else max(q_table, key=q_table.get)
        ix.spread(q_threads)
        best = None
        for _ in range(2):        # first pass warms the pages, keep the faster
            t0 = time.perf_counter()
            ix.query_batch(sk_cpu, threads=q_threads)
            t = time.perf_counter() - t0
            best = t if best is None else min(best, t)
        t_q += best
        n_sub_ix += 1
        for i in range(n_par):
            exp_cols[i, b0:b0 + n_sub] = ix.counts(sk_cpu[par[i]])
        del ix, sub
    leg_done("gather_leg_incl_index_builds")
    parity_counts = bool(np.array_equal(cnt.astype(np.uint32), exp_cols))
    # hit lists of the timed step for these queries: threshold + order of the oracle's columns
    off = hit_off[si].cpu().numpy()
    g_hc, g_hg = hc.cpu().numpy(), hg.cpu().numpy()
    parity_hits = True
    for i in range(n_par):
        gids = np.nonzero(exp_cols[i] >= p.min_score)[0]
        order = np.lexsort((-gids.astype(np.int64), -exp_cols[i, gids].astype(np.int64)))
        lo, hi = int(off[pos[par[i]]]), int(off[pos[par[i]] + 1])
        parity_hits &= bool(np.array_equal(g_hc[lo:hi].astype(np.uint32), exp_cols[i, gids][order]) and
                            np.array_equal(g_hg[lo:hi].astype(np.uint32), gids[order].astype(np.uint32)))
    val = n_s / (t_sk + t_q)

    # ---- the REAL reference's own code beside the port, where oracle/_ref travelled with the repo: its
    # compute_sketch (src/niqki_index.cpp:335-358) on a few of the same genomes, and its insert_sketch + query_sketch
    # (:362-370, :633-687; vector<gid>[2^27] buckets) on a 2048-genome sub-index, each on ONE thread next to the port
    # on one thread -- same work, so the ratio says what the port's flat arrays change
    ref = None
    if po.have_ref():
        try:
            n_r, n_ri, n_rq = 4, 2048, 16
            t0 = time.perf_counter()
            r = po.Ref(K=K, S=S, W=W, H=H, J=J, out_path="/tmp/niqki_bench_ref_%d.gz" % os.getpid())
            t_ctor = time.perf_counter() - t0
            t0 = time.perf_counter()
            r_sk = [r.compute_sketch(rec[i]) for i in range(n_r)]
            t_ref_sk = time.perf_counter() - t0
            t0 = time.perf_counter()
            po.sketch_batch(p, rec[:n_r].reshape(-1), rec_off[:n_r + 1], threads=1)
            t_port_sk = time.perf_counter() - t0
            sub = eng.get_sketches(0, n_ri)
            t0 = time.perf_counter()
            for g in range(n_ri):
                r.insert(sub[g])
            t_ref_ins = time.perf_counter() - t0
            t0 = time.perf_counter()
            ixr = po.Index(p, sub, threads=1)
            t_port_ins = time.perf_counter() - t0
            t0 = time.perf_counter()
            r_hits = [r.query(sk_cpu[i]) for i in range(n_rq)]
            t_ref_q = time.perf_counter() - t0
            t0 = time.perf_counter()
            p_hits = [ixr.query(sk_cpu[i]) for i in range(n_rq)]
            t_port_q = time.perf_counter() - t0
            same = all(np.array_equal(r_sk[i], sk_cpu[i]) for i in range(n_r)) and \
                all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(r_hits, p_hits))
            r.close()
            try:
                os.remove("/tmp/niqki_bench_ref_%d.gz" % os.getpid())
            except OSError:
                pass
            del ixr, sub
            ref = {"kind": "reference", "threads": 1, "constructor_s": t_ctor,
                   "sketch_genomes_per_s": n_r / t_ref_sk, "port_sketch_genomes_per_s": n_r / t_port_sk,
                   "insert_genomes_per_s": n_ri / t_ref_ins, "port_index_build_genomes_per_s": n_ri / t_port_ins,
                   "query_per_s_on_%d_genomes" % n_ri: n_rq / t_ref_q, "port_query_per_s_on_%d_genomes" % n_ri: n_rq / t_port_q,
                   "reference_equals_port": bool(same),
                   "sample": "the reference's own Index class (oracle/_ref/libniqki_ref.so, built from /root/reference/src by "
                             "oracle/Makefile): compute_sketch on %d query genomes; insert_sketch of %d indexed genomes; query_sketch "
                             "of %d queries against them -- one thread, the port timed on the same work beside it" % (n_r, n_ri, n_rq)}
        except Exception as e:      # noqa: BLE001 -- a baseline beside the baseline: never the run's failure
            ref = {"error": str(e)[:200]}
    leg_done("reference_sample")
    # ---- the REAL reference on the job's CPUs over the same sample: its own Index with ALL N genomes inserted (from the
    # sketches the GPU stored -- sketch parity is checked above and in the tests), its compute_sketch and query_sketch
    # driven from `cores` OpenMP threads, one record per thread at a time like its drivers (src/niqki_index.cpp:523-540;
    # oracle/ref_harness.cpp ref_*_batch).  Same queries, same thread count as the port's headline.
    def reference_on_the_whole_index():
        try:
            t0 = time.perf_counter()
            r = po.Ref(K=K, S=S, W=W, H=H, J=J, out_path="/tmp/niqki_bench_ref_all_%d.gz" % os.getpid())
            t_ctor = time.perf_counter() - t0
            t_ins = 0.0
            for b0 in range(0, N, 4096):
                sub = eng.get_sketches(b0, min(4096, N - b0))
                t0 = time.perf_counter()
                r.insert_batch(sub, threads=cores)
                t_ins += time.perf_counter() - t0
                del sub
            t0 = time.perf_counter()
            r_sk = r.sketch_batch(rec.reshape(-1), rec_off, threads=cores)
            t_rsk = time.perf_counter() - t0
            t0 = time.perf_counter()
            r_off, r_hc, r_hg = r.query_batch(r_sk, threads=cores)
            t_rq = time.perf_counter() - t0
            same = bool(np.array_equal(r_sk, sk_gpu))
            for i in range(n_s):     # the reference's own hit lists against the GPU's of the timed step, the whole sample
                lo, hi = int(off[pos[i]]), int(off[pos[i] + 1])
                rl, rh = int(r_off[i]), int(r_off[i + 1])
                same &= bool(np.array_equal(g_hc[lo:hi].astype(np.uint32), r_hc[rl:rh]) and np.array_equal(g_hg[lo:hi].astype(np.uint32), r_hg[rl:rh]))
            r.close()
            try:
                os.remove("/tmp/niqki_bench_ref_all_%d.gz" % os.getpid())
            except OSError:
                pass
            return {"value": n_s / (t_rsk + t_rq), "unit": "genomes/s", "cores": cores, "kind": "reference",
                         "sketch_genomes_per_s": n_s / t_rsk, "query_genomes_per_s": n_s / t_rq,
                         "index_build_s": {"constructor": round(t_ctor, 2), "insert_%d_genomes" % N: round(t_ins, 2)},
                         "gpu_hit_lists_equal_the_references": same,
                         "sample": "%d query genomes (spread evenly over the batch) of step %d through the reference's own Index (oracle/_ref/libniqki_ref.so, built from "
                                   "/root/reference/src by oracle/Makefile) holding all %d genomes: compute_sketch + query_sketch from %d "
                                   "OpenMP threads, one record per thread at a time like src/niqki_index.cpp:523-540 (%.2f + %.2f s)"
                                   % (n_s, si, N, cores, t_rsk, t_rq)}
        except Exception as e:      # noqa: BLE001 -- a baseline beside the baseline: never the run's failure
            return {"error": str(e)[:200]}
    if po.have_ref() and hasattr(po.Ref, "query_batch") and not os.environ.get("NIQKI_BENCH_NO_REFERENCE_INDEX"):
        deferred.append(reference_on_the_whole_index)
    return {
        "value": val, "unit": "genomes/s", "cores": cores, "kind": "port",
        "reference": None,      # (filled in by a later leg while the budget lasts: legs())
        "host_logical_cpus": host["logical_cpus"], "host_physical_cores": host["physical_cores"],
        "host_affinity_cpus": host["affinity_cpus"], "host_cgroup_cpu_quota": host["cgroup_cpu_quota"],
        "threads_tried": cand,
        "sketch_genomes_per_s_by_threads": {str(k): v for k, v in sorted(sk_table.items())},
        "gather_queries_per_s_by_threads_first_sub_index": {str(k): v for k, v in sorted(q_table.items())},
        "gather_threads": q_threads,
        "sample": "%d query genomes of step %d, spread evenly over its %d launch positions: sketch leg timed in full on %d threads (the CPUs this job has: its cgroup quota, else the "
                  "physical cores; %.2f s); gather leg = the oracle's query loop on %d threads timed on each of the %d sub-indexes of "
                  "<= 16384 genomes that together hold all %d genomes, summed (%.2f s); the index arrays first touched by the "
                  "gathering threads" % (n_s, si, per, cores, t_sk, q_threads, n_sub_ix, N, t_q),
        "reference_sample": ref,
        "seconds_by_leg": t_leg,
        "parity": {"sketch_bit_exact": parity_sketch, "counts_bit_exact_all_columns": parity_counts,
                   "hit_lists_bit_exact": parity_hits, "queries_checked": n_par,
                   "launch_positions_checked": [int(pos[i]) for i in par]},
    }


def extra_workloads(niqki_amd, torch, dev, args, budget, out, guarded, reference_leg, no_cpu=False):
    """BASELINE.json configs[1] (1k genomes: index + self query), configs[4] (150-base reads against a
    10k-genome index, S=12 W=10, lines-mode semantics) and the matrix path, each timed with inputs
    resident in HBM and each with an in-run parity check against the oracle; then the host program on files.
    Every leg fills its key of `out` and starts only while the budget has its estimated time left."""
    from oracle import pyoracle as po
    seed, L = args.seed + 1, args.len
    K, S, W, H, J = 31, 15, 12, 4, 0.1
    F = 1 << S
    GB = 250
    p = po.make_params(K, S, W, H, J)
    t32 = lambda a: torch.from_numpy(np.asarray(a).astype(np.int64)).to(torch.int32).to(dev)  # noqa: E731
    t64 = lambda a: torch.from_numpy(np.asarray(a).astype(np.int64)).to(dev)  # noqa: E731

    def timed(fn, eng, reps=1):
        eng.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        eng.synchronize()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    def hits_equal(off, hc_, hg_, i, cols, min_score):
        gids = np.nonzero(cols >= min_score)[0]
        order = np.lexsort((-gids.astype(np.int64), -cols[gids].astype(np.int64)))
        lo, hi = int(off[i]), int(off[i + 1])
        return bool(np.array_equal(hc_[lo:hi].astype(np.uint32), cols[gids][order]) and
                    np.array_equal(hg_[lo:hi].astype(np.uint32), gids[order].astype(np.uint32)))

    def engine(S, W):
        e = niqki_amd.Engine(K=31, S=S, W=W, H=4, J=0.1, device=dev.index)
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        e.set_option("record_len_hint", L)
        return e

    # ---- configs[1]: 1k synthetic 5 Mbp genomes, index + self query, K=31 S=15 W=12 ----
    def leg_configs1():
        N1 = 1000
        fam, mem, rate = genome_spec(np.arange(N1), N1 // 10, 10)
        seq = torch.zeros(N1 * L + niqki_amd.SEQ_PAD, dtype=torch.uint8, device=dev)
        ro = t64(np.arange(N1 + 1, dtype=np.int64) * L)
        sk = torch.empty((N1, F), dtype=torch.int32, device=dev)
        cap = N1 * 64
        ho = torch.zeros(N1 + 1, dtype=torch.int64, device=dev)
        hc = torch.zeros(cap, dtype=torch.int32, device=dev)
        hg = torch.zeros(cap, dtype=torch.int32, device=dev)
        t_index = None
        for attempt in range(3):                     # first pass warms kernels and allocations; the faster of the next two counts
            e = engine(S, W)
            if attempt == 0:
                e.synth_dev(seed, t32(fam), t32(mem), t32(rate), N1, L, L, seq)

            def index_1k():
                e.sketch_dev(seq, ro, N1, sk)
                e.insert_dev(sk, N1)
                e.build()
            t = timed(index_1k, e)
            t_index = t if attempt < 2 else min(t, t_index)   # (a fresh handle's first allocations vary from box to box)
            if attempt < 2:
                e.close()
        e.query_sequences_dev(seq, ro, N1, ho, hc, hg, cap)
        t_query = timed(lambda: e.query_sequences_dev(seq, ro, N1, ho, hc, hg, cap), e, reps=3)
        p = po.make_params(K, S, W, H, J)
        skh = sk.cpu().numpy()
        n_par = 4
        par_sk = all(np.array_equal(skh[i], po.compute_sketch(p, seq[i * L:(i + 1) * L].cpu().numpy())) for i in range(n_par))
        ix = po.Index(p, skh)
        off, c_, g_ = ho.cpu().numpy(), hc.cpu().numpy(), hg.cpu().numpy()
        par_hits = all(hits_equal(off, c_, g_, i, ix.counts(skh[i]), p.min_score) for i in range(0, N1, 16))
        cpu1 = None
        if not no_cpu:
            th = po.lib().nqo_max_threads()
            t0 = time.perf_counter()
            ix.query_batch(skh[:256], threads=th)
            cpu1 = {"gather_genomes_per_s": 256 / (time.perf_counter() - t0), "threads": th, "sample": "256 self queries, gather leg only"}
        out["configs1_1k_index_self_query"] = {
            "workload": "1000 synthetic %d bp genomes (100 families of 10), K=31 S=15 W=12 J=0.1, bases resident in HBM" % L,
            "index_genomes_per_s": N1 / t_index, "index_s": t_index,
            "query_genomes_per_s": N1 / t_query, "query_s": t_query, "hits": int(off[N1]),
            "parity": {"sketch_bit_exact": bool(par_sk), "sketches_checked": n_par, "hit_lists_bit_exact": bool(par_hits),
                       "queries_checked": len(range(0, N1, 16))},
            "cpu_oracle": cpu1,
        }
        del ix
        e.close()

        if not budget.want("sketch_k21", 4):
            return
        # ---- -K other than 31 (src/niqki.cpp:260): the sketch kernel's rate at K = 21 on the same bytes ----
        p21 = po.make_params(21, S, W, H, J)
        e21 = niqki_amd.Engine(K=21, S=S, W=W, H=4, J=0.1, device=dev.index)
        e21.set_stream(torch.cuda.current_stream().cuda_stream)
        e21.set_option("record_len_hint", L)
        e21.sketch_dev(seq, ro, N1, sk)
        t21 = timed(lambda: e21.sketch_dev(seq, ro, N1, sk), e21, reps=2)
        e31 = engine(S, W)
        sk31 = torch.empty_like(sk)
        e31.sketch_dev(seq, ro, N1, sk31)
        t31 = timed(lambda: e31.sketch_dev(seq, ro, N1, sk31), e31, reps=2)
        sk21h = sk[:2].cpu().numpy()
        par21 = all(np.array_equal(sk21h[i], po.compute_sketch(p21, seq[i * L:(i + 1) * L].cpu().numpy())) for i in range(2))
        out["sketch_k21"] = {
            "workload": "the sketch kernel alone on 1000 synthetic %d bp genomes, K = 21 beside K = 31 (S=15 W=12)" % L,
            "k21_gkmers_per_s": N1 * (L - 21) / t21 / 1e9, "k31_gkmers_per_s": N1 * (L - 31) / t31 / 1e9,
            "k21_over_k31": (N1 * (L - 21) / t21) / (N1 * (L - 31) / t31),
            "parity": {"sketch_bit_exact": bool(par21), "sketches_checked": 2},
        }
        e21.close()
        e31.close()
        del sk31, seq, sk

    def leg_matrix():
        # ---- matrix path: all-vs-all of a 10k-genome index (query_range / query_matrix) ----
        NM = 10_000
        e = engine(S, W)
        seq = torch.zeros(GB * L + niqki_amd.SEQ_PAD, dtype=torch.uint8, device=dev)
        skb = torch.empty((GB, F), dtype=torch.int32, device=dev)
        rob = t64(np.arange(GB + 1, dtype=np.int64) * L)
        for g0 in range(0, NM, GB):
            fam, mem, rate = genome_spec(np.arange(g0, g0 + GB), NM // 100, 100)
            e.synth_dev(seed + 1, t32(fam), t32(mem), t32(rate), GB, L, L, seq)
            e.sketch_dev(seq, rob, GB, skb)
            e.insert_dev(skb, GB)
        e.build()
        del skb
        rows = 1024
        mstride = niqki_amd.row_stride(NM)
        mat = torch.zeros((rows, mstride), dtype=torch.int16, device=dev)

        def matrix_all():
            for t0_ in range(0, NM, rows):
                e._ck(e.L.niqki_matrix_range(e.h, t0_, min(NM, t0_ + rows), mat.data_ptr(), mstride, niqki_amd.MEM_DEVICE))
        matrix_all()
        t_mat = timed(matrix_all, e)
        e._ck(e.L.niqki_matrix_range(e.h, 5000, 5000 + rows, mat.data_ptr(), mstride, niqki_amd.MEM_DEVICE))
        e.synchronize()
        mh = mat.cpu().numpy().view(np.uint16)[:, :NM]
        ixm = po.Index(p, e.get_sketches(0, NM))
        exp = ixm.matrix_range(5000, 5008)            # [a][t - begin]
        par_mat = bool(np.array_equal(mh[:8], exp.T))
        out["matrix_10k"] = {
            "workload": "all-vs-all hit matrix of a 10000-genome index (100 families of 100), K=31 S=15 W=12, counters left on the device",
            "rows_per_s": NM / t_mat, "seconds": t_mat,
            "algorithmic_bytes": 4 * NM * F + 2 * NM * NM, "algorithmic_gbs": (4 * NM * F + 2 * NM * NM) / t_mat / 1e9,
            "parity": {"rows_bit_exact_vs_bucket_cooccurrence": par_mat, "rows_checked": 8},
        }
        del ixm, mat
        e.close()

    def leg_configs4():
        # ---- configs[4]: 150-base reads vs a 10k-genome index, K=31 S=12 W=10 (lines mode: one sketch per read) ----
        S4, W4 = 12, 10
        F4, N4, NR, RL, RB = 1 << S4, 10_000, 10_485_760, 150, 65536
        e = engine(S4, W4)
        seq = torch.zeros(GB * L + niqki_amd.SEQ_PAD, dtype=torch.uint8, device=dev)
        rob = t64(np.arange(GB + 1, dtype=np.int64) * L)
        skb = torch.empty((GB, F4), dtype=torch.int32, device=dev)
        for g0 in range(0, N4, GB):
            fam, mem, rate = genome_spec(np.arange(g0, g0 + GB), N4 // 100, 100)
            e.synth_dev(seed + 2, t32(fam), t32(mem), t32(rate), GB, L, L, seq)
            e.sketch_dev(seq, rob, GB, skb)
            e.insert_dev(skb, GB)
        e.build()
        del seq, skb
        # 10 M distinct reads generated on the device: read i = 150 bases at a pseudo-random offset of a
        # pseudo-random indexed genome, 1 % substitutions of its own (164/16384)
        rng = np.random.default_rng(args.seed)
        reads = torch.zeros(NR * RL + niqki_amd.SEQ_PAD, dtype=torch.uint8, device=dev)
        src_g = rng.integers(0, N4, NR)
        src_off = rng.integers(0, L - RL, NR).astype(np.uint64)
        CH = 1 << 20
        for a in range(0, NR, CH):
            gch = src_g[a:a + CH]
            fam, mem, rate = genome_spec(gch, N4 // 100, 100)
            e.synth_reads_dev(seed + 2, t32(fam), t32(mem), t32(rate), t64(src_off[a:a + CH]), t32(np.arange(a, a + len(gch))),
                              164, len(gch), RL, RL, reads[a * RL:])
        e.set_option("record_len_hint", RL)
        rro = t64(np.arange(RB + 1, dtype=np.int64) * RL)
        rsk = torch.empty((RB, F4), dtype=torch.int32, device=dev)
        rstride = niqki_amd.row_stride(N4)
        rcnt = torch.zeros((RB, rstride), dtype=torch.int16, device=dev)
        # the threshold: reads share few slots with 5 Mbp genomes, so J is set where hits exist -- the count
        # the source genome of a read typically reaches (calibrated on the first batch, then fixed)
        e.sketch_dev(reads, rro, RB, rsk)
        e.query_counts_dev(rsk, RB, rcnt, rstride)
        e.synchronize()
        c0 = rcnt[:4096].cpu().numpy().view(np.uint16)[:, :N4]
        own = c0[np.arange(4096), src_g[:4096]]
        min_score = max(2, int(np.percentile(own, 25)))
        del rcnt
        e.set_option("min_score", min_score)
        rcap = RB * 256
        rho = torch.zeros(RB + 1, dtype=torch.int64, device=dev)
        rhc = torch.zeros(rcap, dtype=torch.int32, device=dev)
        rhg = torch.zeros(rcap, dtype=torch.int32, device=dev)
        tots = torch.zeros(NR // RB, dtype=torch.int64, device=dev)

        def all_reads():
            for k, a in enumerate(range(0, NR, RB)):
                e.sketch_dev(reads[a * RL:], rro, RB, rsk)
                e.query_dev(rsk, RB, rho, rhc, rhg, rcap)
                tots[k] = rho[RB]                     # device-side copy: no host sync inside the timed loop
        all_reads()
        e.profile(True)
        e.profile_reset()
        t_reads = timed(all_reads, e)
        kprof = {n_: e.profile_read(kc)[0] for n_, kc in (("sketch", niqki_amd.KC_SKETCH), ("gather", niqki_amd.KC_GATHER), ("hits", niqki_amd.KC_HITS))}
        e.profile(False)
        th_ = tots.cpu().numpy()
        # parity on the last batch: oracle sketches of 64 reads, dense counters and hit lists against an oracle index
        a = NR - RB
        rd = reads[a * RL:(a + 64) * RL].cpu().numpy().reshape(64, RL)
        host_rd = e.synth_reads_host(seed + 2, *genome_spec(src_g[a:a + 64], N4 // 100, 100), src_off[a:a + 64], np.arange(a, a + 64), 164, RL)
        p4 = po.make_params(K, S4, W4, H, J)
        p4.min_score = min_score
        rskh = rsk[:64].cpu().numpy()
        par_sk = all(np.array_equal(rskh[i], po.compute_sketch(p4, rd[i])) for i in range(64))
        # densification passes of these reads (the oracle's loop is the reference's: src/niqki_index.cpp:313-331)
        passes = [po.densify(p4, po.sketch_accumulate(p4, rd[i]))[1] for i in range(64)]
        passes_per_read = float(np.mean([x for x in passes if x > 0])) if any(x > 0 for x in passes) else 0.0
        ix4 = po.Index(p4, e.get_sketches(0, N4))
        off, c_, g_ = rho.cpu().numpy(), rhc.cpu().numpy(), rhg.cpu().numpy()
        par_hits = all(hits_equal(off, c_, g_, i, ix4.counts(rskh[i]), min_score) for i in range(64))
        cpu4 = None
        if not no_cpu:
            th = po.lib().nqo_max_threads()
            n_c = 2048
            rdc = reads[:n_c * RL].cpu().numpy()
            t0 = time.perf_counter()
            skc = po.sketch_batch(p4, rdc, (np.arange(n_c + 1) * RL).astype(np.uint64), threads=th)
            ix4.query_batch(skc, threads=th)
            cpu4 = {"reads_per_s": n_c / (time.perf_counter() - t0), "threads": th, "kind": "port",
                    "sample": "%d reads: sketch (densification dominated, src/niqki_index.cpp:313-331) + query" % n_c}
        # ceilings of the two big kernels of this workload, measured / computed now:
        #  * sketch: its time is the densification passes (one LDS round trip each at 8 wavefronts per CU); the ceiling is
        #    the rate of those passes with nothing but their LDS traffic and exit test (niqki_measure_alu(5))
        #  * gather: HBM -- per read its sketch in (4F), its counter row out (2N) and the table entries and bucket ids
        #    of the look-ups the per-slot class mask lets through (measured: T ids + their entries)
        pass_rate = e.measure_alu(5)
        sk_s = kprof["sketch"] * 1e-3
        sk_pass_rate = NR * passes_per_read / sk_s if sk_s else 0.0
        T4 = float(e.gathered_dev(rsk, RB).sum()) / RB        # ids gathered per read (last batch)
        lists = int(e.stat("last_hits_form")) == 1       # the hits left the gather kernel as ordered lists: no counter rows
        hpr = float(th_.sum()) / NR
        g_bytes = NR * (4 * F4 + 4 * T4 + 8 * T4 + (4 * hpr + 4 if lists else 2 * N4))
        g_s = kprof["gather"] * 1e-3
        h_bytes = NR * ((4 + 12 + 4 * hpr) if lists else 2 * N4) + 8 * float(th_.sum())
        h_s = kprof["hits"] * 1e-3
        ceilings = {
            "sketch": {"bound": "lds round trips of the densification passes", "passes_per_read": passes_per_read,
                       "achieved_passes_per_s": sk_pass_rate, "peak_passes_per_s": pass_rate,
                       "frac": sk_pass_rate / pass_rate if pass_rate else None,
                       "note": "peak = niqki_measure_alu(5): the pass loop of the short-read kernel (9 one-wave workgroups per CU, two "
                               "proposals + two read-backs per lane and pass) with nothing but its LDS traffic and exit test, measured "
                               "in this run; achieved = reads/s of the sketch kernel x the passes the oracle's serial loop takes for "
                               "64 of these reads (k-mer hashing and the entry list are inside the kernel's time, outside the peak; "
                               "the kernel's own passes read their targets a window ahead and its last 16 cells are filled in closed "
                               "form, so it issues fewer LDS instructions than the probe's passes: the fraction can pass 1)"},
            "gather": {"bound": "hbm", "algorithmic_bytes_per_read": g_bytes / NR, "gathered_ids_per_read": T4,
                       "achieved": g_bytes / g_s / 1e9 if g_s else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": g_bytes / g_s / 1e9 / HBM_PEAK_GBS if g_s else None,
                       "class_mask": int(e.stat("class_mask")),
                       "hit_lists": lists,
                       "note": "per read: its 2^S-cell sketch in, the entries and ids of the buckets it touches, and its ordered hit list "
                               "out (4 bytes per hit; a 2N-byte counter row only for a read with more than hit_list_cap hits, or with "
                               "option hit_lists = 0); with the per-slot class mask the 2^S table look-ups of a read are not memory "
                               "traffic any more.  The kernel is bound by the latency of a read's dependent steps at 5 workgroups per "
                               "CU, not by these bytes"},
            "hits": {"bound": "launch latency of three small kernels (sizes -> offsets, lists -> places, overflowing lists ordered)",
                     "bytes_per_read": h_bytes / NR, "achieved": h_bytes / h_s / 1e9 if h_s else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": h_bytes / h_s / 1e9 / HBM_PEAK_GBS if h_s else None,
                     "ms_per_batch": kprof["hits"] / (NR // RB)},
        }
        out["configs4_reads_vs_10k_index"] = {
            "workload": "%d distinct 150-base reads (1 %% substitutions, generated on the device) against a 10000-genome index, "
                        "K=31 S=12 W=10, one sketch per read, batches of %d resident in HBM" % (NR, RB),
            "reads_per_s": NR / t_reads, "seconds": t_reads, "min_score": min_score, "J_equivalent": min_score / F4,
            "hits_total": int(th_.sum()), "hits_per_read": float(th_.sum()) / NR, "hit_overflow": bool((th_ > rcap).any()),
            "kernel_ms": {k_: round(v, 1) for k_, v in kprof.items()},
            "ceilings": ceilings,
            "parity": {"device_reads_equal_host_generator": bool(np.array_equal(rd, host_rd)), "sketch_bit_exact": bool(par_sk),
                       "hit_lists_bit_exact": bool(par_hits), "reads_checked": 64},
            "cpu_oracle": cpu4,
        }
        e.close()

    # ---- the `niqki` host program on FILES: FASTA bytes in the page cache -> hits in a gz ----
    # (tools/bench_cli.py in child processes: whole-file mode, plain and gzip level 1 inputs; rates from the
    # program's own phase clocks, its start-up reported beside them)
    def child_json(name, cmd, own_timeout, estimate, scratch_dir=None):
        """the last stdout line of a child process as JSON, or None (dropped / killed / failed); scratch_dir goes either way"""
        import shutil
        if os.environ.get("NIQKI_BENCH_TEST_HANG") == name:      # (CPU / GPU rehearsal of a child that never returns)
            cmd = [sys.executable, "-c", "import time; time.sleep(100000)"]
        try:
            r = budget.child(name, cmd, own_timeout, estimate)
            if r is None or r[0] != 0:
                if r is not None:
                    budget.dropped.append({"leg": name, "exit_code": r[0]})
                return None
            return json.loads(r[1].decode().strip().splitlines()[-1])
        except (ValueError, IndexError):
            return None
        finally:
            if scratch_dir:
                shutil.rmtree(scratch_dir, ignore_errors=True)
            budget.lap(name)

    cli = {}

    def leg_cli(tag, n_files, gz, estimate):
        scratch = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", "niqki_bench_cli_%d" % os.getpid())
        cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_cli.py"), "--genomes", str(n_files), "--len", str(L), "--dir", scratch]
        if gz:
            # gzip -6 files (gzip's default), crossing PCIe as they are and inflated on the device (nq_inflate.hip); the
            # same run once more with every file inflated by the reader threads (NIQKI_HOST_NO_GPU_INFLATE=1)
            cmd += ["--gz", "--host-inflate-too"]
        else:
            cmd += ["--reference", "128", "--reads", "4000000"]
        j = child_json("cli_files." + tag, cmd, 600, estimate, scratch)
        if j:
            cli[tag] = {"files": n_files, "index_genomes_per_s": j["index_genomes_per_s"], "query_genomes_per_s": j["query_genomes_per_s"],
                        "index_file_GBps": j["index_fasta_GBps"], "process_startup_s": j["startup_s"],
                        # the -I phase of the FIRST process that reads the freshly written files (tmpfs: slower for any
                        # reader, `cat` included) beside the rate above, which is the -I phase of the `-I .. -Q ..` run
                        "index_first_pass_genomes_per_s": j.get("index_first_pass_genomes_per_s"),
                        "packed_fasta": "plain FASTA files travel as 2 bits per base in full A/C/G/T lines (niqki_pack_fasta, "
                                        "made by the reader threads while they read); the device restores the files' bytes" if not gz else None,
                        "query_phase_split_s": j.get("query_phase_split_s")}
            if gz:
                cli[tag]["gzip_level"] = j.get("gz_level")
                cli[tag]["file_GB"] = round(j.get("file_bytes", 0) / 1e9, 3)
                cli[tag]["device_inflate"] = ("gzip files cross PCIe as they lie on disk, one wavefront per file inflates them "
                                              "(equal batches of at most 2048 files; CRC-32 and sizes checked on the device)")
                cli[tag]["reader_threads_inflate_instead"] = j.get("host_inflate")
            if j.get("reads_per_s"):
                # BASELINE configs[4] as FILES: `niqki -I fof -l reads.fa -S 12 -W 10` (--querylines: one entry per record),
                # 4 M 150-base reads in a FASTA file in the page cache, the lines phase's own clock
                cli[tag]["lines_mode_reads_per_s"] = j["reads_per_s"]
            if j.get("reference_program"):
                # the reference's OWN program on the first files: its CPU path, and the same binary with its three
                # operators bound to the C ABI (oracle/ref_gpu_ops.cpp) -- what INTEGRATION.md's minimal patch gives
                cli[tag]["reference_program"] = j["reference_program"]
        out["cli_files"] = {"workload": "niqki -I fof -Q fof -J 0.1 on 5 Mbp FASTA files (70 columns) in the page cache, whole-file mode",
                            **cli} if cli else None

    # ---- `niqki -D` / `-L` end to end (tools/bench_dump_cli.py): 8192 genomes, 1.6 GB of buckets ----
    def leg_dump_load():
        j = child_json("dump_load_cli", [sys.executable, os.path.join(ROOT, "tools", "bench_dump_cli.py"), "--genomes", "8192", "--len", "200000"],
                       300, 10)
        if j:
            out["dump_load_cli"] = {
                "workload": "niqki -I fof -D dump.gz, then niqki -L dump.gz -Q q: 8192 synthetic genomes of 200 kbp (K=31 S=15 W=12), the dump a "
                            "file of size-tagged gzip -1 members written and read side by side by the host's threads",
                "dump_file_GB": j["dump_file_GB"], "dump_s": j["dump_s"], "load_s": j["load_s"],
                "hits_after_load_equal_hits_after_index": j["same_hits"],
                "dump_phase": [l for l in j["timing"].get("index_dump", []) if "dump:" in l][-1:]}
    # ---- the device inflate alone: 1024 gzip -6 genome files resident in HBM, one launch (tools/bench_inflate.py) ----
    def leg_inflate():
        j = child_json("gzip_inflate", [sys.executable, os.path.join(ROOT, "tools", "bench_inflate.py"), "--files", "2048", "--len", str(L),
                                        "--distinct", "8", "--reps", "2"], 300, 15)
        if j:
            out["gzip_inflate"] = {
                "workload": "2048 gzip -6 FASTA files of %d bp inflated in one launch of nq::inflate_kernel (one wavefront per file, eight per "
                            "CU: each file's last 8 KB of window in LDS, matches that reach further back read from its flushed output), "
                            "bytes checked against the files' own" % L,
                "files_in_flight_by_kernel_form": j.get("files_in_flight"),
                "kernel_ms": j["kernel_ms"], "files_per_s": j["files_per_s"], "inflated_GBps": j["raw_GBps"], "compressed_GBps": j["wire_GBps"],
                "per_file": j["per_file"],
                "bound": "the latency of one wavefront's dependent instructions (a DEFLATE stream is serial): every file takes the whole "
                         "launch, throughput = files in flight / that time",
                "hbm_frac": round((j["raw_GBps"] + j["wire_GBps"]) / 8000.0, 4),
                "cycles_per_file_round": round(j["kernel_ms"] * 1e-3 * 2.4e9 / max(j["per_file"]["rounds"], 1)),
                "note": "bytes written + read over the 8 TB/s peak: a hundredth -- the kernel is nowhere near memory; a round (64 bit offsets "
                        "decoded at once, their tokens walked, <= 64 bytes written) is ~190 instructions of one wavefront, two wavefronts per "
                        "SIMD (profiles/r05_inflate_phase_clocks.txt); cycles_per_file_round at a nominal 2.4 GHz",
                "zlib_one_host_thread_files_per_s": j["zlib_one_thread_files_per_s"]}
    # ---- the legs in the order of what they are worth; each starts only while its estimated time is left ----
    for name, est, fn in (("configs4_reads_vs_10k_index", 14, leg_configs4), ("configs1_1k_index_self_query", 10, leg_configs1),
                          ("matrix_10k", 8, leg_matrix)):
        if budget.want(name, est):
            guarded(name, fn)
    reference_leg()
    for tag, n_files, gz, est in (("plain_fasta", 2048, False, 32), ("gzip_fasta", 2048, True, 58)):
        leg_cli(tag, n_files, gz, est)
    leg_dump_load()
    leg_inflate()


if __name__ == "__main__":
    main()
