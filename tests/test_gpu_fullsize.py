"""GPU parity at BASELINE.json's full size: configs[2] = 100 000 synthetic 5 Mbp genomes indexed,
K=31 S=15 W=12 H=4 J=0.1, queried with the bench's own query genomes.

The oracle cannot sketch 500 Gbp, so the test is anchored in two steps, both bit-exact:
  * sketches: a sample of the index genomes and every checked query genome is re-sketched by
    the oracle from the same synthetic bytes (host generator == device generator);
  * index + counting loop + threshold + order (src/niqki_index.cpp:652-666, :685): the stored
    sketches are read back in seven blocks of <= 16 384 genomes, each block becomes an oracle
    index (the oracle's own insert + CSR), and the oracle's counters of ALL 100 000 columns are
    compared with niqki_query_counts; the oracle's thresholded, ordered hit lists built from
    those columns are compared with niqki_query's.

The checked queries are every 61st of the 4096 of the bench's first batch (68 rows), so that the launch form
bench.py times -- ONE call of 4096 queries: four 1024-query groups of lookup_rows_kernel, the locality order
over all 4096 -- is compared with the oracle at launch positions of every group, and so is the
`--shard-of 8 --batch 32768` form (8 ranks x 4096 queries through the sparse exchange).
"""
import numpy as np
import pytest

import bench

pytestmark = pytest.mark.gpu

N, L, FAMILY, SEED = 100_000, 5_000_000, 100, 20261003
K, S, W, H, J = 31, 15, 12, 4, 0.1
F = 1 << S
NQ_BIG = 4096      # queries of one call: bench.py's --batch, the launch form it times (streamed table rows + order)
CHECK = np.arange(0, NQ_BIG, 61)   # launch positions compared with the oracle: 17 in each 1024-query group of the pre-pass
NQ = len(CHECK)    # 68 checked queries (dense columns + hit lists)
NQ_SKETCH = 6      # queries re-sketched by the oracle (0.1 s of CPU each)
BLOCK = 16384


BIG = {}   # the NQ_BIG device-resident query sketches of the fixture below


@pytest.fixture(scope="module")
def big(native):
    """The bench's index and its first query batch, built exactly as bench.py builds them."""
    import torch
    dev = torch.device("cuda")
    e = native.Engine(K=K, S=S, W=W, H=H, J=J)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.set_option("record_len_hint", L)
    e.reserve(N)
    n_fam = N // FAMILY
    GB = 256
    t32 = lambda a: torch.from_numpy(a.astype(np.int64)).to(torch.int32).to(dev)  # noqa: E731
    seq = torch.zeros(GB * L + native.SEQ_PAD, dtype=torch.uint8, device=dev)
    sk = torch.empty((GB, F), dtype=torch.int32, device=dev)
    ro = torch.from_numpy(np.arange(GB + 1, dtype=np.int64) * L).to(dev)
    for g0 in range(0, N, GB):
        n = min(GB, N - g0)
        fam, mem, rate = bench.genome_spec(np.arange(g0, g0 + n), n_fam, FAMILY)
        e.synth_dev(SEED, t32(fam), t32(mem), t32(rate), n, L, L, seq)
        e.sketch_dev(seq, ro if n == GB else torch.from_numpy(np.arange(n + 1, dtype=np.int64) * L).to(dev), n, sk)
        e.insert_dev(sk, n)
    e.build()
    # queries 0..NQ_BIG-1 of the bench's first batch (every 10th from a family that is not indexed), sketched in
    # blocks of GB; rows CHECK of them are the ones checked column by column
    qfam, qmem, qrate = bench.query_spec(np.arange(NQ_BIG), n_fam)
    qsk_big = torch.empty((NQ_BIG, F), dtype=torch.int32, device=dev)
    qseq = None
    for q0 in range(0, NQ_BIG, GB):
        e.synth_dev(SEED, t32(qfam[q0:q0 + GB]), t32(qmem[q0:q0 + GB]), t32(qrate[q0:q0 + GB]), GB, L, L, seq)
        e.sketch_dev(seq, ro, GB, qsk_big[q0:q0 + GB])
        if q0 == 0:
            e.synchronize()
            qseq = seq[:NQ_SKETCH * L].cpu().numpy().reshape(NQ_SKETCH, L).copy()
    e.synchronize()
    del seq
    BIG["qsk"] = qsk_big
    yield e, qsk_big[torch.from_numpy(CHECK).to(dev)].cpu().numpy(), qseq, (qfam[CHECK], qmem[CHECK], qrate[CHECK])
    BIG.clear()
    e.close()


def test_config3_sketches_of_index_and_queries_vs_oracle(native, po, big):
    e, _, qseq, _ = big
    p = po.make_params(K, S, W, H, J)
    first = BIG["qsk"][:NQ_SKETCH].cpu().numpy()
    for i in range(NQ_SKETCH):
        assert np.array_equal(first[i], po.compute_sketch(p, qseq[i])), i
    # indexed genomes: first, last, a family ancestor and a 5 % member, from the host generator
    n_fam = N // FAMILY
    for g in (0, 4_299, 65_471, N - 1):
        fam, mem, rate = bench.genome_spec(np.array([g]), n_fam, FAMILY)
        seq = native.synth_genome_host(SEED, int(fam[0]), int(mem[0]), int(rate[0]), L)
        assert np.array_equal(e.get_sketches(g, 1)[0], po.compute_sketch(p, seq)), g


def test_config3_all_columns_and_hit_lists_vs_oracle(native, po, big):
    e, qsk, _, (qfam, _, _) = big
    assert e.n_genomes == N and e.tile_genomes() < N          # two counter tiles, striped
    p = po.make_params(K, S, W, H, J)
    assert p.min_score == 3276 == e.min_score
    got = e.query_counts(qsk)                                   # [NQ][N] u16 through the C ABI
    exp = np.zeros((NQ, N), np.uint32)
    for b0 in range(0, N, BLOCK):
        n = min(BLOCK, N - b0)
        sub = e.get_sketches(b0, n)
        ix = po.Index(p, sub)
        for q in range(NQ):
            exp[q, b0:b0 + n] = ix.counts(qsk[q])
        del ix, sub
    assert np.array_equal(got.astype(np.uint32), exp)
    # the same through the slot-major look-up pre-pass (what a 4096-query launch uses by default)
    e.set_option("lookup_prepass", 1)
    assert np.array_equal(e.query_counts(qsk), got)
    e.set_option("lookup_prepass", -1)
    # the launch form bench.py times -- 4096 queries in one call: table rows streamed through LDS
    # (lookup_rows_kernel, four groups of 1024 queries), locality probe + order over all 4096,
    # gather_kernel<1024, 32, -1, 0, true>: rows CHECK of that call (every 61st launch position, all four
    # groups) are the oracle's columns, the hit lists likewise
    import torch
    big_sk = BIG["qsk"]
    dev = big_sk.device
    d_check = torch.from_numpy(CHECK).to(dev)
    stride = native.row_stride(N)
    d_counts = torch.zeros((NQ_BIG, stride), dtype=torch.int16, device=dev)
    e.query_counts_dev(big_sk, NQ_BIG, d_counts, stride)
    e.synchronize()
    assert e.stat("last_gather_form") == 7                      # pre-pass, its streamed-rows kernel, locality order
    big_rows = d_counts[d_check][:, :N].cpu().numpy().view(np.uint16)
    assert np.array_equal(big_rows.astype(np.uint32), exp)
    for grp_ in range(NQ_BIG // 1024):
        assert ((CHECK // 1024) == grp_).sum() >= 16
    # ... and rows further back in the batch equal what a small call (look-ups inside the gather kernel) gives
    tail = slice(NQ_BIG - 48, NQ_BIG)
    small = e.query_counts(big_sk[tail].cpu().numpy())
    assert e.stat("last_gather_form") & 3 == 0
    assert np.array_equal(d_counts[tail, :N].cpu().numpy().view(np.uint16), small)
    cap_big = NQ_BIG * 256
    b_off = torch.zeros(NQ_BIG + 1, dtype=torch.int64, device=dev)
    b_hc, b_hg = torch.zeros(cap_big, dtype=torch.int32, device=dev), torch.zeros(cap_big, dtype=torch.int32, device=dev)
    e.query_dev(big_sk, NQ_BIG, b_off, b_hc, b_hg, cap_big)
    e.synchronize()
    assert e.stat("last_gather_form") == 7
    b_off, b_hc, b_hg = b_off.cpu().numpy(), b_hc.cpu().numpy().astype(np.uint32), b_hg.cpu().numpy().astype(np.uint32)
    assert int(b_off[NQ_BIG]) <= cap_big
    del d_counts
    # threshold + order from the oracle's columns: greater<pair<count, gid>>, :662-666, :685 -- the small call's
    # lists AND the lists at launch positions CHECK of the 4096-query call
    off, hc, hg = e.query(qsk)
    n_with_hits = 0
    for q in range(NQ):
        gids = np.nonzero(exp[q] >= p.min_score)[0]
        order = np.lexsort((-gids.astype(np.int64), -exp[q, gids].astype(np.int64)))
        ec, eg = exp[q, gids][order], gids[order].astype(np.uint32)
        lo, hi = int(off[q]), int(off[q + 1])
        assert np.array_equal(hc[lo:hi], ec) and np.array_equal(hg[lo:hi], eg), q
        lo, hi = int(b_off[CHECK[q]]), int(b_off[CHECK[q] + 1])
        assert np.array_equal(b_hc[lo:hi], ec) and np.array_equal(b_hg[lo:hi], eg), ("launch position", int(CHECK[q]))
        n_with_hits += len(gids) > 0
        if qfam[q] < N // FAMILY:                              # mutant of an indexed family: hits stay inside it
            assert len(gids) > 0 and (gids // FAMILY == qfam[q]).all(), q
        else:
            assert len(gids) == 0, q
    assert n_with_hits >= NQ * 8 // 10
    # the oracle's own threshold/sort (nqo_hits_from_counts) agrees with the numpy restatement
    L_ = po.lib()
    BIG["hits"] = (b_off, b_hc, b_hg)                         # (checked against the oracle at rows CHECK just above)
    row = np.ascontiguousarray(exp[0])
    oc, og = np.empty(N, np.uint32), np.empty(N, np.uint32)
    k = L_.nqo_hits_from_counts(row.ctypes.data, N, p.min_score, oc.ctypes.data, og.ctypes.data, N)
    assert np.array_equal(oc[:k], hc[int(off[0]):int(off[1])]) and np.array_equal(og[:k], hg[int(off[0]):int(off[1])])


def test_candidates_from_counts_vs_numpy(native):
    """niqki_candidates_from_counts (the sparse multi-GPU exchange's first step): ids with
    counter >= threshold, any order, -1 padding, n_cand exact also when the list overflows."""
    import torch
    dev = torch.device("cuda")
    rng = np.random.default_rng(12)
    e = native.Engine(K=31, S=8, W=8, H=4)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    for n_gids, stride, cap, thr in ((1000, 1000, 64, 40), (70001, 70002, 256, 3), (513, 600, 8, 1), (64, 64, 64, 0),
                                     (300, 300, 16, 65535)):
        nq = 7
        c = rng.integers(0, 60, (nq, stride)).astype(np.uint16)
        c[0, :] = 0                       # no candidate
        c[1, :n_gids] = thr               # every genome exactly at the threshold (overflows cap)
        c[2, :n_gids] = max(thr, 1) - 1   # every genome just below it
        c[3, n_gids:] = 65535             # columns beyond n_gids never count
        c[4, 0] = c[4, n_gids - 1] = 65535
        d_c = torch.from_numpy(c.view(np.int16)).to(dev)
        cand = torch.full((nq, cap), 7, dtype=torch.int32, device=dev)
        ncand = torch.zeros(nq, dtype=torch.int32, device=dev)
        e.candidates_dev(d_c, nq, stride, n_gids, thr, cap, cand, ncand)
        e.synchronize()
        cand, ncand = cand.cpu().numpy(), ncand.cpu().numpy()
        for q in range(nq):
            want = np.nonzero(c[q, :n_gids] >= thr)[0]
            assert ncand[q] == len(want), (q, n_gids, thr)
            k = min(len(want), cap)
            assert (cand[q, k:] == -1).all()
            got = cand[q, :k]
            assert len(set(got.tolist())) == k and set(got.tolist()) <= set(want.tolist())
            if len(want) <= cap:
                assert sorted(got.tolist()) == want.tolist()
    e.close()


def test_config4_eight_slot_shards_at_full_size(native, big):
    """BASELINE.json configs[3] on the one GPU a test box has: the 100 000-genome index cut into 8
    slot shards (niqki_group_*, all ranks in this process, device copies standing in for RCCL), the
    bench's queries through the sparse exchange and through the dense one: the hit lists of the
    whole-range index (checked against the oracle above)."""
    import torch
    dev = torch.device("cuda")
    e, qsk, _, _ = big
    G, per = 8, 8
    qsk = qsk[:G * per]
    shards = []
    for r in range(G):
        b, s_end = native.group_slot_range(r, G, S)
        sh = native.Engine(K=K, S=S, W=W, H=H, J=J, slot_begin=b, slot_end=s_end)
        sh.set_stream(torch.cuda.current_stream().cuda_stream)
        sh.reserve(N)
        shards.append(sh)
    grp = native.Group(shards)
    # the stored sketches of the whole-range index, through the group's slice exchange
    INS = 512
    for g0 in range(0, N, G * INS):
        n = min(G * INS, N - g0)
        blk = torch.full((G * INS, F), -1, dtype=torch.int32, device=dev)
        blk[:n] = torch.from_numpy(e.get_sketches(g0, n)).to(dev)
        grp.insert_dev([blk[r * INS:(r + 1) * INS] for r in range(G)], INS, n)
    del blk
    assert all(sh.n_genomes == N for sh in shards)
    off, hc, hg = e.query(qsk)
    dq = torch.from_numpy(qsk).to(dev)
    loc = [dq[r * per:(r + 1) * per].contiguous() for r in range(G)]
    for mode in (1, 2):                       # sparse, dense
        grp.set_option("exchange", mode)
        res = grp.query(loc, per)
        for r in range(G):
            o, c, g_ = res[r]
            for i in range(per):
                q = r * per + i
                lo, hi = int(off[q]), int(off[q + 1])
                assert np.array_equal(c[int(o[i]):int(o[i + 1])], hc[lo:hi]) and np.array_equal(g_[int(o[i]):int(o[i + 1])], hg[lo:hi]), (mode, q)
    # the form `bench.py --shard-of 8 --batch 32768` times: every rank brings 4096 queries, each shard's gather sees
    # 32 768 query slices and leaves candidates + survivors (no counter rows), the sparse exchange makes the hits.
    # Rank r's queries are the bench batch rotated by 517 r positions, so every rank's slice holds every checked
    # query at another launch position; each of the 8 x 4096 lists must be the whole-range call's list of that
    # query (compared with the oracle at rows CHECK in the test above, or right here if that test did not run).
    big_sk = BIG["qsk"]
    if "hits" in BIG:
        w_off, w_hc, w_hg = BIG["hits"]
    else:
        cap_big = NQ_BIG * 256
        t_off = torch.zeros(NQ_BIG + 1, dtype=torch.int64, device=dev)
        t_hc, t_hg = torch.zeros(cap_big, dtype=torch.int32, device=dev), torch.zeros(cap_big, dtype=torch.int32, device=dev)
        e.query_dev(big_sk, NQ_BIG, t_off, t_hc, t_hg, cap_big)
        e.synchronize()
        w_off, w_hc, w_hg = t_off.cpu().numpy(), t_hc.cpu().numpy().astype(np.uint32), t_hg.cpu().numpy().astype(np.uint32)
        for i, q in enumerate(CHECK[:G * per]):
            lo, hi = int(off[i]), int(off[i + 1])
            assert np.array_equal(w_hc[int(w_off[q]):int(w_off[q + 1])], hc[lo:hi]) and np.array_equal(w_hg[int(w_off[q]):int(w_off[q + 1])], hg[lo:hi])
    SHIFT = 517
    loc_big = [torch.roll(big_sk, -SHIFT * r, 0).contiguous() for r in range(G)]      # row i of rank r = query (i + 517 r) mod 4096
    grp.set_option("exchange", 1)
    res = grp.query(loc_big, NQ_BIG, capacity=NQ_BIG * 256)
    w_len = np.diff(w_off)
    for r in range(G):
        o, c, g_ = res[r]
        src = (np.arange(NQ_BIG) + SHIFT * r) % NQ_BIG
        assert np.array_equal(np.diff(o.astype(np.int64)), w_len[src]), r
        # the rank's lists back in the whole-range call's query order: one comparison per rank
        order = np.argsort(src, kind="stable")
        take = np.concatenate([np.arange(int(o[i]), int(o[i + 1])) for i in order]) if int(o[NQ_BIG]) else np.zeros(0, np.int64)
        assert np.array_equal(c[take], w_hc[:int(w_off[NQ_BIG])]) and np.array_equal(g_[take], w_hg[:int(w_off[NQ_BIG])]), r
    del loc_big
    assert grp.stat("overflows") == 0
    grp.close()
    for sh in shards:
        sh.close()


@pytest.mark.parametrize("Kx", [21, 27])
def test_full_size_sketches_other_k(native, po, Kx):
    """5 Mbp genomes of the bench's generator at -K 21 / 27 (src/niqki.cpp:260, src/niqki_index.cpp:225-236), S = 15
    W = 12: the fast filtered loop with a run-time K, against the oracle on the same bytes."""
    p = po.make_params(Kx, S, W, H, J)
    e = native.Engine(K=Kx, S=S, W=W, H=H, J=J)
    n_fam = N // FAMILY
    gids = (0, 1, 99)
    spec = [bench.genome_spec(np.array([g]), n_fam, FAMILY) for g in gids]
    seqs = [native.synth_genome_host(SEED, int(f[0]), int(m[0]), int(r[0]), L) for f, m, r in spec]
    sk = e.sketch(seqs)
    for i in range(len(gids)):
        assert np.array_equal(sk[i], po.compute_sketch(p, seqs[i])), (Kx, gids[i])
    e.close()
