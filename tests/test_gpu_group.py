"""GPU: a slot-sharded index behind the C ABI (niqki_group_*, niqki_amd/csrc/nq_group.hip).

World-size G groups are run on ONE GPU: all ranks in this process, the shards sharing the device,
so device-to-device copies stand in for the RCCL collectives while every exchange kernel, the
slot-shard gather, the candidate logic and the per-rank threshold run as they do on G GPUs.  The
answers must be those of one whole-range handle (itself checked against the oracle elsewhere) and
of the oracle.  A world-1 group goes through librccl itself."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_data(S, W, N, nq, seed):
    rng = np.random.default_rng(seed)
    F = 1 << S
    fam = rng.integers(0, 1 << W, (12, F)).astype(np.int32)
    sk = fam[rng.integers(0, 12, N)].copy()
    noise = rng.random((N, F)) < 0.4
    sk[noise] = rng.integers(0, 1 << W, int(noise.sum()))
    sk[rng.random((N, F)) < 0.02] = -1
    sk[5] = sk[3]
    q = fam[rng.integers(0, 12, nq)].copy()
    m = rng.random((nq, F)) < 0.25
    q[m] = rng.integers(0, 1 << W, int(m.sum()))
    q[1] = -1                                   # no hit at all
    q[2, ::2] = 1 << W                          # cells outside [0, 2^W) are never queried
    return sk, q


def shards_and_group(native, world, S, W, min_score, **kw):
    engines = []
    for r in range(world):
        b, e = native.group_slot_range(r, world, S)
        engines.append(native.Engine(K=31, S=S, W=W, H=3, min_score_value=min_score, slot_begin=b, slot_end=e))
    return engines, native.Group(engines, **kw)


def run_group(native, torch, engines, grp, sk, q, per):
    """Inserts sk through the group in ragged batches, queries q; returns per-query (counts, gids)."""
    dev = torch.device("cuda")
    world, F = len(engines), sk.shape[1]
    for e in engines:
        e.set_stream(torch.cuda.current_stream().cuda_stream)
    # insert: batches of world * ins_per rows, the last one ragged
    ins_per = 37
    for a in range(0, sk.shape[0], world * ins_per):
        blk = sk[a:a + world * ins_per]
        pad = np.full((world * ins_per, F), -1, np.int32)
        pad[:blk.shape[0]] = blk
        loc = [torch.from_numpy(pad[r * ins_per:(r + 1) * ins_per].copy()).to(dev) for r in range(world)]
        grp.insert_dev(loc, ins_per, blk.shape[0])
    assert all(e.n_genomes == sk.shape[0] for e in engines)
    nq = q.shape[0]
    pad = np.full((world * per, F), -1, np.int32)
    pad[:nq] = q
    loc = [torch.from_numpy(pad[r * per:(r + 1) * per].copy()).to(dev) for r in range(world)]
    res = grp.query(loc, per, capacity=8)         # forces the capacity retry
    out = []
    for r in range(world):
        off, hc, hg = res[r]
        for i in range(per):
            out.append((hc[int(off[i]):int(off[i + 1])], hg[int(off[i]):int(off[i + 1])]))
    return out[:nq]


@pytest.mark.parametrize("world,exchange", [(8, "sparse"), (8, "dense"), (8, "overflow"), (3, "sparse"), (2, "dense"), (1, "sparse")])
def test_emulated_slot_shards_equal_the_whole_index(native, po, world, exchange):
    import torch
    S, W, N, NQ, MS = 9, 8, 1234, 21, 40
    sk, q = make_data(S, W, N, NQ, 5 + world)
    whole = native.Engine(K=31, S=S, W=W, H=3, min_score_value=MS)
    whole.insert(sk)
    w_off, w_hc, w_hg = whole.query(q)
    engines, grp = shards_and_group(native, world, S, W, MS)
    assert grp.stat("rccl") == (1 if world == 1 else 0)     # shards on one device: local transport
    grp.set_option("exchange", 2 if exchange == "dense" else 1)
    if exchange == "overflow":
        grp.set_option("cand_cap", 2)                        # lists overflow -> the step is redone densely
    per = -(-NQ // world)
    # what the group will do is what the pure plan function says (the function nq_group.hip calls per batch, and the
    # one the CPU protocol test over gloo takes its decisions from)
    plan = native.group_plan(world, S, MS, 2 if exchange == "dense" else 1, per, N, 2 if exchange == "overflow" else 256)
    assert grp.stat("sparse") == plan.sparse and plan.cand_threshold == -(-MS // world) and plan.slice_slots == -(-(1 << S) // world)
    got = run_group(native, torch, engines, grp, sk, q, per)
    p = po.make_params(31, S, W, 3, 0.0)
    p.min_score = MS
    ix = po.Index(p, sk)
    for i in range(NQ):
        lo, hi = int(w_off[i]), int(w_off[i + 1])
        assert np.array_equal(got[i][0], w_hc[lo:hi]) and np.array_equal(got[i][1], w_hg[lo:hi]), (world, exchange, i)
        ehc, ehg = ix.query(q[i], min_score=MS)
        assert np.array_equal(got[i][0], ehc) and np.array_equal(got[i][1], ehg), i
    assert sum(len(g[0]) for g in got) > 50
    assert (grp.stat("overflows") >= 1) == (exchange == "overflow")   # every (retried) step of the overflowing case
    # the shards' partial hit vectors add up to the whole index's
    tot = sum(e.query_counts(q).astype(np.uint32) for e in engines)
    assert np.array_equal(tot, whole.query_counts(q).astype(np.uint32))
    grp.close()
    for e in engines + [whole]:
        e.close()


def test_group_rejects_wrong_shards(native):
    S = 8
    a = native.Engine(K=31, S=S, W=8, H=3, slot_begin=0, slot_end=128)
    b = native.Engine(K=31, S=S, W=8, H=3, slot_begin=100, slot_end=256)      # not rank 1's range
    with pytest.raises(native.NiqkiError) as ei:
        native.Group([a, b])
    assert ei.value.code == 1 and "slots [128, 256)" in str(ei.value)
    a.close()
    b.close()


def test_rccl_world1_group(native, po):
    """One rank, transport = librccl (all-to-all, all-gather and reduce-scatter on a 1-rank
    communicator): the C ABI's RCCL path end to end on the one GPU a test box has."""
    import torch
    S, W, N, NQ, MS = 10, 8, 900, 16, 50
    sk, q = make_data(S, W, N, NQ, 77)
    e = native.Engine(K=31, S=S, W=W, H=3, min_score_value=MS)
    grp = native.Group([e], first_rank=0, world=1, group_id=native.group_new_id())
    assert grp.stat("rccl") == 1 and grp.stat("sparse") == 1
    got = run_group(native, torch, [e], grp, sk, q, NQ)
    p = po.make_params(31, S, W, 3, 0.0)
    p.min_score = MS
    ix = po.Index(p, sk)
    for i in range(NQ):
        ehc, ehg = ix.query(q[i], min_score=MS)
        assert np.array_equal(got[i][0], ehc) and np.array_equal(got[i][1], ehg), i
    # dense exchange through RCCL on the same index
    grp.set_option("exchange", 2)
    dev = torch.device("cuda")
    res = grp.query([torch.from_numpy(q.copy()).to(dev)], NQ)
    off, hc, hg = res[0]
    for i in range(NQ):
        assert np.array_equal(hc[int(off[i]):int(off[i + 1])], got[i][0]) and np.array_equal(hg[int(off[i]):int(off[i + 1])], got[i][1])
    grp.close()
    e.close()


def test_row_free_sparse_steps_vs_dense_counters(native):
    """niqki_query_survivors / niqki_survivor_counts / niqki_hits_from_candidates -- the sparse exchange without
    counter rows, step by step on a slot shard -- against the shard's dense counters (niqki_query_counts)."""
    import torch
    dev = torch.device("cuda")
    S, W, N, NQ, MS = 10, 8, 3000, 37, 60
    sk, q = make_data(S, W, N, NQ, 99)
    F = 1 << S
    for sb, se in ((0, F), (F // 4, F // 2)):                       # whole range, and one shard of four
        e = native.Engine(K=31, S=S, W=W, H=3, min_score_value=MS, slot_begin=sb, slot_end=se)
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        e.insert(sk)
        dense = e.query_counts(q).astype(np.uint32)                 # [NQ][N]
        thr, sthr, C, SC = (MS, MS // 2, 64, 512) if sb == 0 else (MS // 4, MS // 8, 64, 1024)
        dq = torch.from_numpy(q).to(dev)
        cand = torch.zeros((NQ, C), dtype=torch.int32, device=dev)
        ncand = torch.zeros(NQ, dtype=torch.int32, device=dev)
        surv = torch.zeros((NQ, SC, 2), dtype=torch.int32, device=dev)
        nsurv = torch.zeros(NQ, dtype=torch.int32, device=dev)
        e.query_survivors_dev(dq, NQ, thr, sthr, C, SC, cand, ncand, surv, nsurv)
        e.synchronize()
        h_cand, h_nc, h_surv, h_ns = cand.cpu().numpy(), ncand.cpu().numpy(), surv.cpu().numpy(), nsurv.cpu().numpy()
        for i in range(NQ):
            want_c = np.nonzero(dense[i] >= thr)[0]
            want_s = np.nonzero(dense[i] >= sthr)[0]
            assert h_nc[i] == len(want_c) and h_ns[i] == len(want_s), i
            if len(want_c) <= C:
                assert sorted(h_cand[i, :len(want_c)].tolist()) == want_c.tolist() and (h_cand[i, len(want_c):] == -1).all()
            if len(want_s) <= SC:
                got = h_surv[i, :len(want_s)]
                order = np.argsort(got[:, 0])
                assert np.array_equal(got[order, 0], want_s) and np.array_equal(got[order, 1].astype(np.uint32), dense[i, want_s])
        assert (h_ns <= SC).all() and h_ns.max() > 0
        # any ids -- survivors, genomes that are not (counted from the sketch store), -1, duplicates
        rng = np.random.default_rng(3)
        m = 96
        ids = rng.integers(0, N, (NQ, m)).astype(np.int32)
        for i in range(NQ):
            k = min(int(h_ns[i]), SC, 40)
            ids[i, :k] = h_surv[i, :k, 0]
        ids[:, 50:54] = -1
        ids[:, 60] = ids[:, 0]
        d_ids = torch.from_numpy(ids).to(dev)
        out = torch.zeros((NQ, m), dtype=torch.int16, device=dev)
        e.survivor_counts_dev(dq, NQ, d_ids, m, surv, nsurv, SC, out)
        e.synchronize()
        got = out.cpu().numpy().view(np.uint16).astype(np.uint32)
        exp = np.where(ids >= 0, np.take_along_axis(dense, np.maximum(ids, 0).astype(np.int64), axis=1), 0)
        assert np.array_equal(got, exp)
        # hits from candidates: every genome that reaches min_score is among the ids (plus weaker ones, -1s, duplicates)
        m2 = 128
        ids2 = np.full((NQ, m2), -1, np.int32)
        for i in range(NQ):
            strong = np.nonzero(dense[i] >= MS)[0][:100]
            ids2[i, :len(strong)] = strong[::-1]
            ids2[i, 100:110] = rng.integers(0, N, 10)
            if len(strong):
                ids2[i, 120] = strong[0]
        tot = np.where(ids2 >= 0, np.take_along_axis(dense, np.maximum(ids2, 0).astype(np.int64), axis=1), 0).astype(np.uint16)
        cap = NQ * m2
        h_off = torch.zeros(NQ + 1, dtype=torch.int64, device=dev)
        hc, hg = torch.zeros(cap, dtype=torch.int32, device=dev), torch.zeros(cap, dtype=torch.int32, device=dev)
        e.hits_from_candidates_dev(torch.from_numpy(ids2).to(dev), torch.from_numpy(tot.view(np.int16)).to(dev), NQ, m2, h_off, hc, hg, cap)
        e.synchronize()
        off, c_, g_ = h_off.cpu().numpy(), hc.cpu().numpy().astype(np.uint32), hg.cpu().numpy().astype(np.uint32)
        n_hits = 0
        for i in range(NQ):
            gids = np.unique(ids2[i][ids2[i] >= 0])                 # distinct ids offered ...
            gids = gids[dense[i, gids] >= MS]                       # ... whose sum reaches min_score
            order = np.lexsort((-gids.astype(np.int64), -dense[i, gids].astype(np.int64)))
            lo, hi = int(off[i]), int(off[i + 1])
            assert np.array_equal(c_[lo:hi], dense[i, gids][order]) and np.array_equal(g_[lo:hi], gids[order].astype(np.uint32)), i
            n_hits += hi - lo
        assert n_hits > (20 if sb == 0 else 0)
        e.close()
