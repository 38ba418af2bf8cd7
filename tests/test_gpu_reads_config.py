"""GPU parity for BASELINE.json configs[4], the short-sequence path (--indexlines / --querylines semantics:
one sketch per record): a 10 000-genome index at K=31 S=12 W=10 built with bench.py's generator, 65 536
device-generated 150-base reads through niqki_query (sketch_reads_kernel with its densification passes,
gather + hits at S=12), and for 96 of them -- spread over the batch -- the oracle's sketches, dense
counters and thresholded, ordered hit lists (src/niqki_index.cpp:412-430 -> :633-687)."""
import numpy as np
import pytest

import bench

pytestmark = pytest.mark.gpu

K, S, W, H = 31, 12, 10, 4
F, N, L = 1 << S, 10_000, 5_000_000
NR, RL, SEED = 65_536, 150, 20261003 + 3
CHECK = np.r_[0:32, 30_000:30_032, NR - 32:NR]      # reads compared with the oracle


def test_config5_reads_vs_10k_index(native, po, monkeypatch):
    import torch
    dev = torch.device("cuda")
    t32 = lambda a: torch.from_numpy(np.asarray(a).astype(np.int64)).to(torch.int32).to(dev)  # noqa: E731
    t64 = lambda a: torch.from_numpy(np.asarray(a).astype(np.int64)).to(dev)  # noqa: E731
    e = native.Engine(K=K, S=S, W=W, H=H, J=0.1)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.set_option("record_len_hint", L)
    GB = 250
    seq = torch.zeros(GB * L + native.SEQ_PAD, dtype=torch.uint8, device=dev)
    ro = t64(np.arange(GB + 1, dtype=np.int64) * L)
    skb = torch.empty((GB, F), dtype=torch.int32, device=dev)
    for g0 in range(0, N, GB):
        fam, mem, rate = bench.genome_spec(np.arange(g0, g0 + GB), N // 100, 100)
        e.synth_dev(SEED, t32(fam), t32(mem), t32(rate), GB, L, L, seq)
        e.sketch_dev(seq, ro, GB, skb)
        e.insert_dev(skb, GB)
    e.build()
    del seq, skb
    assert e.n_genomes == N
    # reads: 150 bases at a pseudo-random offset of a pseudo-random indexed genome, 1 % substitutions of their own
    rng = np.random.default_rng(5)
    src_g = rng.integers(0, N, NR)
    src_off = rng.integers(0, L - RL, NR).astype(np.uint64)
    reads = torch.zeros(NR * RL + native.SEQ_PAD, dtype=torch.uint8, device=dev)
    fam, mem, rate = bench.genome_spec(src_g, N // 100, 100)
    e.synth_reads_dev(SEED, t32(fam), t32(mem), t32(rate), t64(src_off), t32(np.arange(NR)), 164, NR, RL, RL, reads)
    e.set_option("record_len_hint", RL)
    rro = t64(np.arange(NR + 1, dtype=np.int64) * RL)
    # the threshold where hits exist for reads (they share few slots with 5 Mbp genomes): fixed, low
    min_score = 2
    e.set_option("min_score", min_score)
    cap = NR * 512
    d_off = torch.zeros(NR + 1, dtype=torch.int64, device=dev)
    d_hc, d_hg = torch.zeros(cap, dtype=torch.int32, device=dev), torch.zeros(cap, dtype=torch.int32, device=dev)
    e.query_sequences_dev(reads, rro, NR, d_off, d_hc, d_hg, cap)       # niqki_query_sequences: sketch + query
    e.synchronize()
    off = d_off.cpu().numpy()
    assert 0 < int(off[NR]) <= cap
    hc, hg = d_hc[:int(off[NR])].cpu().numpy().astype(np.uint32), d_hg[:int(off[NR])].cpu().numpy().astype(np.uint32)
    # the device's reads are the host generator's
    rd = reads[:NR * RL].cpu().numpy().reshape(NR, RL)
    host_rd = e.synth_reads_host(SEED, fam[CHECK], mem[CHECK], rate[CHECK], src_off[CHECK], CHECK, 164, RL)
    assert np.array_equal(rd[CHECK], host_rd)
    # oracle: sketches of the checked reads, its own index over the stored sketches, counters, threshold, order
    p = po.make_params(K, S, W, H, 0.1)
    p.min_score = min_score
    ix = po.Index(p, e.get_sketches(0, N))
    d_sk = torch.empty((len(CHECK), F), dtype=torch.int32, device=dev)
    sub = torch.from_numpy(np.ascontiguousarray(rd[CHECK]).reshape(-1)).to(dev)
    sub = torch.cat([sub, torch.zeros(native.SEQ_PAD, dtype=torch.uint8, device=dev)])
    e.sketch_dev(sub, t64(np.arange(len(CHECK) + 1, dtype=np.int64) * RL), len(CHECK), d_sk)
    e.synchronize()
    sk = d_sk.cpu().numpy()
    counts = e.query_counts(sk)
    n_hit_reads = 0
    for j, i in enumerate(CHECK):
        exp_sk = po.compute_sketch(p, rd[i])
        assert np.array_equal(sk[j], exp_sk), i                          # densification passes included
        cols = ix.counts(exp_sk)
        assert np.array_equal(counts[j].astype(np.uint32), cols), i
        gids = np.nonzero(cols >= min_score)[0]
        order = np.lexsort((-gids.astype(np.int64), -cols[gids].astype(np.int64)))
        lo, hi = int(off[i]), int(off[i + 1])
        assert np.array_equal(hc[lo:hi], cols[gids][order]) and np.array_equal(hg[lo:hi], gids[order].astype(np.uint32)), i
        n_hit_reads += hi > lo
        if hi > lo:
            assert int(src_g[i]) // 100 in set((hg[lo:hi] // 100).tolist()) or cols[src_g[i]] < min_score
    assert n_hit_reads >= len(CHECK) // 4
    # The hits left the gather kernel as ordered lists (single small tile: no counter row per read) ...
    assert e.stat("last_hits_form") == 1
    nh = int(off[NR])
    per_read = np.diff(off)
    assert per_read.max() > 256, "the batch should hold reads whose lists overflow the default capacity"

    def again():
        o = torch.zeros(NR + 1, dtype=torch.int64, device=dev)
        c, g = torch.zeros(cap, dtype=torch.int32, device=dev), torch.zeros(cap, dtype=torch.int32, device=dev)
        e.query_sequences_dev(reads, rro, NR, o, c, g, cap)
        e.synchronize()
        return o, c, g
    # ... with a list capacity that nearly every read with hits overflows (forced fall-back to the counter row), with the
    # largest capacity, and in the form that writes and re-reads every counter row: the same bytes
    for key, val, form in (("hit_list_cap", 4, 1), ("hit_list_cap", 2048, 1), ("hit_lists", 0, 0)):
        e.set_option(key, val)
        o, c, g = again()
        assert e.stat("last_hits_form") == form
        assert torch.equal(o, d_off) and torch.equal(c[:nh], d_hc[:nh]) and torch.equal(g[:nh], d_hg[:nh]), (key, val)
    e.set_option("hit_lists", 1)
    e.set_option("hit_list_cap", 256)
    # a capacity below the total: offsets exact, the hits that fit are the first ones
    small = int(off[NR // 2]) + 5
    o = torch.zeros(NR + 1, dtype=torch.int64, device=dev)
    c, g = torch.zeros(small, dtype=torch.int32, device=dev), torch.zeros(small, dtype=torch.int32, device=dev)
    e.query_sequences_dev(reads, rro, NR, o, c, g, small)
    e.synchronize()
    whole = int(off[NR // 2])          # (queries that end within the capacity are complete)
    assert torch.equal(o, d_off) and torch.equal(c[:whole], d_hc[:whole]) and torch.equal(g[:whole], d_hg[:whole])
    # the per-slot class mask (single-tile index) was in use ...
    assert e.stat("class_mask") == 1
    # ... and an index built without it answers the same bytes (the mask only skips look-ups of empty buckets)
    monkeypatch.setenv("NIQKI_HMASK", "0")
    e.set_option("tile_genomes", 0)        # (forces a rebuild)
    e.build()
    assert e.stat("class_mask") == 0
    d_off2 = torch.zeros(NR + 1, dtype=torch.int64, device=dev)
    d_hc2, d_hg2 = torch.zeros(cap, dtype=torch.int32, device=dev), torch.zeros(cap, dtype=torch.int32, device=dev)
    e.query_sequences_dev(reads, rro, NR, d_off2, d_hc2, d_hg2, cap)
    e.synchronize()
    nh = int(off[NR])
    assert torch.equal(d_off2, d_off) and torch.equal(d_hc2[:nh], d_hc[:nh]) and torch.equal(d_hg2[:nh], d_hg[:nh])
    e.close()
