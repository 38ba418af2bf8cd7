"""GPU: niqki_sketch_ahead / niqki_query_ahead (include/niqki_hip.h) -- Index::query_sequence
(src/niqki_index.cpp:691-695) in two halves so that batch i + 1's sketch kernel runs on the handle's sketch lane
beside batch i's gather and hit kernels.  The overlap may never change a result: every batch's hit lists and
sketches are those of the oracle, whatever is in flight beside them."""
import numpy as np
import pytest

from conftest import family_spec

pytestmark = pytest.mark.gpu

K, S, W, H, J = 31, 12, 12, 4, 0.1
F = 1 << S
L = 60_000


def make_index(native, po, torch, n_fam=6, n_mem=8):
    fam, mem, rate = family_spec(n_fam, n_mem)
    genomes = [native.synth_genome_host(11, int(f), int(m), int(r), L) for f, m, r in zip(fam, mem, rate)]
    e = native.Engine(K=K, S=S, W=W, H=H, J=J)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    p = po.make_params(K, S, W, H, J)
    sk = np.stack([po.compute_sketch(p, g) for g in genomes])
    e.insert(sk)
    return e, p, po.Index(p, sk)


def batch(native, torch, b, nq):
    """query batch b: mutants of the indexed families and strangers, as one device buffer + offsets"""
    seqs = [native.synth_genome_host(11, (b * 5 + i) % 8, 1000 + b * 64 + i, 100 + 37 * i, L - 17 * i) for i in range(nq)]
    off = np.zeros(nq + 1, np.int64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    buf = np.concatenate(seqs + [np.zeros(native.SEQ_PAD, np.uint8)])
    return seqs, torch.from_numpy(buf).cuda(), torch.from_numpy(off).cuda()


def expected(po, p, ix, seqs):
    sk = [po.compute_sketch(p, s) for s in seqs]
    hits = [ix.query(s) for s in sk]
    off = np.zeros(len(seqs) + 1, np.int64)
    off[1:] = np.cumsum([len(h[0]) for h in hits])
    return np.stack(sk), off, np.concatenate([h[0] for h in hits]), np.concatenate([h[1] for h in hits])


def test_one_batch_ahead_equals_the_oracle(native, po):
    import torch
    e, p, ix = make_index(native, po, torch)
    nq, nb = 24, 6
    bs = [batch(native, torch, b, nq) for b in range(nb)]
    cap = nq * 64
    outs = []
    e.sketch_ahead_dev(bs[0][1], bs[0][2], nq)
    for b in range(nb):
        if b + 1 < nb:
            e.sketch_ahead_dev(bs[b + 1][1], bs[b + 1][2], nq)
        ho = torch.zeros(nq + 1, dtype=torch.int64, device="cuda")
        hc, hg = torch.zeros(cap, dtype=torch.int32, device="cuda"), torch.zeros(cap, dtype=torch.int32, device="cuda")
        sk = torch.zeros((nq, F), dtype=torch.int32, device="cuda") if b % 2 == 0 else None
        assert e.query_ahead_dev(ho, hc, hg, cap, sk) == nq
        outs.append((ho, hc, hg, sk))
    e.synchronize()
    for b in range(nb):
        esk, eoff, ec, eg = expected(po, p, ix, bs[b][0])
        ho, hc, hg, sk = outs[b]
        n = int(eoff[-1])
        assert np.array_equal(ho.cpu().numpy(), eoff), b
        assert np.array_equal(hc.cpu().numpy()[:n].astype(np.uint32), ec) and np.array_equal(hg.cpu().numpy()[:n].astype(np.uint32), eg), b
        if sk is not None:
            assert np.array_equal(sk.cpu().numpy(), esk), b
    assert n > 0
    e.close()


def test_two_ahead_host_outputs_and_state_errors(native, po):
    import torch
    e, p, ix = make_index(native, po, torch)
    with pytest.raises(native.NiqkiError) as ei:
        e.query_ahead(4)
    assert ei.value.code == 5                                   # NIQKI_E_STATE: nothing sketched ahead
    bs = [batch(native, torch, 10 + b, 8 + b) for b in range(3)]
    e.sketch_ahead_dev(bs[0][1], bs[0][2], 8)
    e.sketch_ahead_dev(bs[1][1], bs[1][2], 9)
    with pytest.raises(native.NiqkiError) as ei:
        e.sketch_ahead_dev(bs[2][1], bs[2][2], 10)
    assert ei.value.code == 5                                   # two batches ahead already
    with pytest.raises(native.NiqkiError) as ei:                # host records are not taken ahead
        e._ck(e.L.niqki_sketch_ahead(e.h, bs[0][0][0].ctypes.data, np.array([0, L], np.uint64).ctypes.data, 1, None, 1, native.MEM_HOST))
    assert ei.value.code == 1
    # the older batch first; a capacity that is too small leaves it the oldest one (the wrapper asks again)
    for b, nq in ((0, 8), (1, 9)):
        esk, eoff, ec, eg = expected(po, p, ix, bs[b][0])
        off, hc, hg, sk = e.query_ahead(nq, capacity=1, want_sketches=True)
        assert np.array_equal(off.astype(np.int64), eoff) and np.array_equal(hc, ec) and np.array_equal(hg, eg), b
        assert np.array_equal(sk, esk), b
        if b == 0:     # ... and other calls on the handle in between see nothing of the lane: a genome more in the index
            g = native.synth_genome_host(11, 5, 3000, 50, L)
            gs = po.compute_sketch(p, g)
            assert np.array_equal(e.sketch([g])[0], gs)
    e.sketch_ahead_dev(bs[2][1], bs[2][2], 10)
    esk, eoff, ec, eg = expected(po, p, ix, bs[2][0])
    off, hc, hg = e.query_ahead(10)
    assert np.array_equal(off.astype(np.int64), eoff) and np.array_equal(hc, ec) and np.array_equal(hg, eg)
    e.close()


def test_reads_ahead(native, po):
    """short records: the one-wavefront sketch kernel (with its fused densification) on the sketch lane, hit lists out"""
    import torch
    S2, W2 = 10, 10
    p = po.make_params(K, S2, W2, H, 0.0)
    p.min_score = 2
    e = native.Engine(K=K, S=S2, W=W2, H=H, min_score_value=2)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    genomes = [native.synth_genome_host(5, f, 0, 0, 40_000) for f in range(12)]
    gsk = np.stack([po.compute_sketch(p, g) for g in genomes])
    e.insert(gsk)
    ix = po.Index(p, gsk)
    rng = np.random.default_rng(8)
    for rep in range(3):
        nr = 300
        reads = [genomes[int(rng.integers(12))][o:o + 150].copy() for o in rng.integers(0, 39_000, nr)]
        off = np.zeros(nr + 1, np.int64)
        off[1:] = np.cumsum([len(r) for r in reads])
        d_seq = torch.from_numpy(np.concatenate(reads + [np.zeros(native.SEQ_PAD, np.uint8)])).cuda()
        d_off = torch.from_numpy(off).cuda()
        e.sketch_ahead_dev(d_seq, d_off, nr)
        got_off, hc, hg, sk = e.query_ahead(nr, want_sketches=True)
        for i in range(0, nr, 7):
            s = po.compute_sketch(p, reads[i])
            assert np.array_equal(sk[i], s), i
            ec, eg = ix.query(s)
            lo, hi = int(got_off[i]), int(got_off[i + 1])
            assert np.array_equal(hc[lo:hi], ec) and np.array_equal(hg[lo:hi], eg), i
    e.close()


@pytest.mark.parametrize("how", ["paged", "s16", "two_tiles_and_delta"])
def test_ahead_on_the_other_index_forms(native, po, how):
    """niqki_query_ahead is niqki_query behind a sketch made ahead: a paged index (counters added page by page), S = 16
    (two counter planes, counts up to 2^16) and a two-tile index with a delta segment answer through it as they do
    through niqki_query_sequences."""
    import torch
    Sx, Wx, Hx = (16, 10, 4) if how == "s16" else (9, 10, 3)
    kw = dict(K=31, S=Sx, W=Wx, H=Hx, J=0.05)
    if how == "paged":
        kw["resident_mib"] = 1
    if how == "two_tiles_and_delta":
        kw["tile_genomes"] = 4160
    e = native.Engine(**kw)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    n_gen, Lg = (40, 90_000) if how == "s16" else ((9000, 3_000) if how == "two_tiles_and_delta" else (300, 20_000))
    fam, mem, rate = family_spec(max(1, n_gen // 10), 10)
    fam, mem, rate = fam[:n_gen], mem[:n_gen], rate[:n_gen]
    genomes = [native.synth_genome_host(23, int(f), int(m), int(r), Lg) for f, m, r in zip(fam, mem, rate)]
    first = n_gen - (600 if how == "two_tiles_and_delta" else 0)
    e.insert(e.sketch(genomes[:first]))
    if how == "two_tiles_and_delta":
        e.build()
        e.insert(e.sketch(genomes[first:]))      # genomes after a build: the delta segment
    nq = 12
    qs = [native.synth_genome_host(23, i % 7, 2000 + i, 90, Lg - 11 * i) for i in range(nq)]
    off = np.zeros(nq + 1, np.int64)
    off[1:] = np.cumsum([len(s) for s in qs])
    d_seq = torch.from_numpy(np.concatenate(qs + [np.zeros(native.SEQ_PAD, np.uint8)])).cuda()
    d_off = torch.from_numpy(off).cuda()
    want = e.query_sequences(qs)
    for rep in range(2):
        e.sketch_ahead_dev(d_seq, d_off, nq)
        got = e.query_ahead(nq)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]), (how, rep)
    assert int(want[0][-1]) > 0
    e.close()


def test_sketch_lane_on_a_part_of_the_device(native, po):
    """option "sketch_lane_cus": the sketch lane made anew on a compute-unit mask; same sketches, same hits"""
    import torch
    e, p, ix = make_index(native, po, torch)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    with pytest.raises(native.NiqkiError):
        e.set_option("sketch_lane_cus", cus + 1)
    seqs, d_seq, d_off = batch(native, torch, 3, 16)
    esk, eoff, ec, eg = expected(po, p, ix, seqs)
    for lane in (0, cus // 4, cus - 8, 0):        # (set between batches: the lane is drained and made anew)
        e.set_option("sketch_lane_cus", lane)
        e.sketch_ahead_dev(d_seq, d_off, 16)
        off, hc, hg, sk = e.query_ahead(16, want_sketches=True)
        assert np.array_equal(sk, esk) and np.array_equal(off, eoff), lane
        assert np.array_equal(hc, ec) and np.array_equal(hg, eg), lane
    e.close()
