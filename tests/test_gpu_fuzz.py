"""GPU: randomized differential test of the whole path against the oracle over the
parameter space the ABI accepts (K 1..31, S 1..15, W <= 15, H <= W, S+W <= 30):
sketch (incl. densification and the inputs on which the reference never returns),
insert, dense counters, thresholded + ordered hits, matrix, dump bytes, -G."""
import numpy as np
import pytest

import os

pytestmark = pytest.mark.gpu
# NIQKI_FUZZ_SCALE=k runs k times as many seeds (one-off campaigns; the default stays small)
SCALE = int(os.environ.get("NIQKI_FUZZ_SCALE", "1"))


def random_record(rng, L, dirty):
    s = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)].copy()
    if dirty and L:
        k = max(1, L // 40)
        s[rng.integers(0, L, k)] = np.frombuffer(b"NnacgtRY-*\r", np.uint8)[rng.integers(0, 11, k)]
    return s


def mutate(rng, s, rate):
    t = s.copy()
    m = rng.random(t.size) < rate
    t[m] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(m.sum()))]
    return t


@pytest.mark.parametrize("seed", range(24 * SCALE))
def test_random_parameters_vs_oracle(native, po, seed, monkeypatch):
    # odd seeds: table look-ups by the slot-major pre-pass wherever the shape allows (S >= 3)
    monkeypatch.setenv("NIQKI_LOOKUP_PREPASS", "1" if seed % 2 else "0")
    rng = np.random.default_rng(1000 + seed)
    K = int(rng.integers(1, 32)) if seed % 3 else 31
    S = int(rng.integers(1, 13))
    W = int(rng.integers(1, min(15, 22 - S) + 1))
    H = int(rng.integers(0, min(W, 6) + 1))
    J = float(rng.choice([0.0, 0.05, 0.3, 0.9]))
    p = po.make_params(K, S, W, H, J)
    e = native.Engine(K=K, S=S, W=W, H=H, J=J)
    # genomes: a few families so that hits exist, lengths from "shorter than K" upwards
    base = [random_record(rng, int(rng.integers(max(K, 40), 6000)), dirty=seed % 2 == 1) for _ in range(3)]
    genomes = []
    for b in base:
        genomes.append(b)
        for r in (0.002, 0.02, 0.1):
            genomes.append(mutate(rng, b, r))
    genomes += [random_record(rng, L, False) for L in (0, K - 1 if K > 1 else 0, K, K + 1, K + 2, 64)]
    exp_sk = np.stack([po.densify(p, po.sketch_accumulate(p, g))[0] for g in genomes])
    sk = e.sketch(genomes)
    assert np.array_equal(sk, exp_sk), (K, S, W, H)
    e.insert(sk)
    ix = po.Index(p, exp_sk)
    queries = np.stack([exp_sk[i] for i in (0, 1, 5, 9, len(genomes) - 1)] +
                       [po.densify(p, po.sketch_accumulate(p, mutate(rng, base[0], 0.05)))[0]])
    cnt = e.query_counts(queries)
    off, hc, hg = e.query(queries)
    for q in range(queries.shape[0]):
        assert np.array_equal(cnt[q].astype(np.uint32), ix.counts(queries[q])), (q, K, S, W, H)
        ec, eg = ix.query(queries[q])
        assert np.array_equal(hc[off[q]:off[q + 1]], ec) and np.array_equal(hg[off[q]:off[q + 1]], eg)
    n = len(genomes)
    assert np.array_equal(e.matrix_range(0, n), ix.matrix_range(0, n).T)
    assert e.export_dump() == ix.dump_bytes()
    e.close()


@pytest.mark.parametrize("seed", range(8 * SCALE))
def test_select_best_h_random(native, po, seed):
    """-G with random constructor parameters: the stale-constant fingerprints of the sketches
    and everything downstream equal the oracle's (itself pinned on the reference)."""
    rng = np.random.default_rng(2000 + seed)
    K = int(rng.integers(8, 32))
    S = int(rng.integers(2, 11))
    W = int(rng.integers(6, min(15, 22 - S) + 1))
    H = int(rng.integers(0, 6))
    G = float(rng.choice([1.0, 150.0, 3e4, 5e6, 1e9]))
    p = po.make_params(K, S, W, H, 0.1, genome_size=G)
    e = native.Engine(K=K, S=S, W=W, H=H, J=0.1)
    assert e.select_best_H(G) == p.H
    base = random_record(rng, 4000, False)
    genomes = [base] + [mutate(rng, base, r) for r in (0.01, 0.05, 0.2)] + [random_record(rng, 900, True)]
    exp_sk = np.stack([po.densify(p, po.sketch_accumulate(p, g))[0] for g in genomes])
    sk = e.sketch(genomes)
    assert np.array_equal(sk, exp_sk), (K, S, W, H, G, p.H)
    e.insert(sk)
    ix = po.Index(p, exp_sk)
    off, hc, hg = e.query(exp_sk)
    for q in range(len(genomes)):
        ec, eg = ix.query(exp_sk[q])
        assert np.array_equal(hc[off[q]:off[q + 1]], ec) and np.array_equal(hg[off[q]:off[q + 1]], eg)
    assert e.export_dump() == ix.dump_bytes()
    e.close()


@pytest.mark.parametrize("seed", range(8 * SCALE))
def test_long_records_random_parameters(native, po, seed):
    """The filtered long-record sketch path (candidate filter + exact re-run) over random K (17..31 take the fast loop,
    below the generic one), S, W, H: a 300 kbp record with dirty stretches, and the same bases as three records of one
    sketch (whole-file mode) -- against the oracle."""
    rng = np.random.default_rng(7000 + seed)
    K = int(rng.integers(17, 32)) if seed % 4 else int(rng.integers(5, 17))
    S = int(rng.integers(8, 13))
    W = int(rng.integers(6, 13))
    H = int(rng.integers(2, min(W, 5) + 1))
    p = po.make_params(K, S, W, H, 0.0)
    e = native.Engine(K=K, S=S, W=W, H=H)
    L = int(rng.integers(250_000, 350_000))
    g = random_record(rng, L, dirty=False)
    for _ in range(3):                                   # a few dirty stretches and single dirty bytes
        a = int(rng.integers(0, L - 200))
        g[a:a + int(rng.integers(1, 120))] = ord("N")
    g[rng.integers(0, L, 20)] = np.frombuffer(b"acgtnRY-", np.uint8)[rng.integers(0, 8, 20)]
    sk = e.sketch([g])
    assert np.array_equal(sk[0], po.compute_sketch(p, g)), (K, S, W, H)
    cuts = sorted(int(x) for x in rng.integers(K + 1, L - K - 1, 2))
    parts = [g[:cuts[0]], g[cuts[0]:cuts[1]], g[cuts[1]:]]
    sk2 = e.sketch(parts, entry_rec=np.array([0, 3], np.uint32))
    acc = np.full(1 << S, -1, np.int32)
    for part in parts:
        po.sketch_accumulate(p, part, acc)
    assert np.array_equal(sk2[0], po.densify(p, acc)[0]), (K, S, W, H)
    e.close()


@pytest.mark.parametrize("seed", range(8 * SCALE))
def test_short_reads_random_parameters(native, po, seed):
    """The one-wavefront short-record kernel (entries in registers, targets read a window ahead, the last cells in
    closed form with ties handed back to the passes) over random K, S, W, H: a few hundred reads of 40..400 bases, some
    of them with foreign bytes or lower case, some repeated (duplicate fingerprints tie in every pass) -- every
    sketch against the oracle's serial loop (src/niqki_index.cpp:313-331)."""
    rng = np.random.default_rng(9000 + seed)
    K = int(rng.integers(15, 32)) if seed % 3 else 31
    S = int(rng.integers(8, 13)) if seed % 4 else int(rng.integers(13, 16))   # (every fourth seed: 2^13 .. 2^15 cells, one or two sketches per CU)
    W = int(rng.integers(4, 15))
    H = int(rng.integers(0, min(W, 6) + 1))
    p = po.make_params(K, S, W, H, 0.0)
    e = native.Engine(K=K, S=S, W=W, H=H)
    reads = []
    for i in range(320 if S <= 12 else 40):
        L = int(rng.integers(40, 401))
        r = random_record(rng, L, dirty=(i % 7 == 0))
        if i % 11 == 0:                       # a short period: few distinct k-mers, many cells to fill from few values
            unit = r[:int(rng.integers(3, 40))]
            r = np.tile(unit, L // unit.size + 1)[:L].copy()
        reads.append(r)
    sk = e.sketch(reads)
    off = np.zeros(len(reads) + 1, np.uint64)
    off[1:] = np.cumsum([r.size for r in reads])
    exp = po.sketch_batch(p, np.concatenate(reads), off)
    bad = [i for i in range(len(reads)) if not np.array_equal(sk[i], exp[i])]
    assert not bad, (K, S, W, H, bad[:8], [reads[i].size for i in bad[:8]])
    e.close()


@pytest.mark.parametrize("seed", range(4 * SCALE))
def test_short_batches_with_long_outliers(native, po, seed):
    """A batch whose average record is short (the one-wavefront kernel takes its 192-entry list, nine sketches per CU)
    but which holds records far longer than that: more occupied cells than the list holds.  Those are flagged by the
    first launch and sketched by the second, with the 384-entry list; records beyond that one too (the plain pass over
    all cells).  Every sketch against the oracle."""
    rng = np.random.default_rng(9500 + seed)
    K, S, W, H = (31, 12, 10, 4) if seed % 2 == 0 else (int(rng.integers(15, 32)), int(rng.integers(9, 13)), int(rng.integers(6, 13)), 3)
    p = po.make_params(K, S, W, H, 0.0)
    e = native.Engine(K=K, S=S, W=W, H=H)
    reads = []
    for i in range(400):
        L = int(rng.integers(60, 160))
        if i % 9 == 0:
            L = int(rng.integers(230, 420))        # between the two lists' capacities
        if i % 67 == 0:
            L = int(rng.integers(500, 900))        # beyond both
        reads.append(random_record(rng, L, dirty=(i % 13 == 0)))
    assert sum(r.size for r in reads) / len(reads) <= 200
    sk = e.sketch(reads)
    off = np.zeros(len(reads) + 1, np.uint64)
    off[1:] = np.cumsum([r.size for r in reads])
    exp = po.sketch_batch(p, np.concatenate(reads), off)
    bad = [i for i in range(len(reads)) if not np.array_equal(sk[i], exp[i])]
    assert not bad, (K, S, W, H, bad[:8], [reads[i].size for i in bad[:8]])
    # the same records one sketch per FILE of several records (entry_rec: cells accumulate over an entry's records)
    er = np.arange(0, len(reads) + 1, 4, dtype=np.uint32)
    sk2 = e.sketch(reads, entry_rec=er)
    for j in range(len(er) - 1):
        acc = np.full(1 << S, -1, np.int32)
        for r in reads[er[j]:er[j + 1]]:
            po.sketch_accumulate(p, r, acc)
        assert np.array_equal(sk2[j], po.densify(p, acc)[0]), (K, S, W, H, j)
    e.close()


@pytest.mark.parametrize("seed", range(3 * SCALE))
def test_mid_length_records_at_the_default_sketch_size(native, po, seed):
    """Records of 0.5 .. 12 kbp at S = 15 with W = 11 / 12 (the reference's defaults): the workgroup kernel stores their
    cells as they are and the distinct-value densification runs as a launch of its own, a thread's hash words in its
    registers (the three tables do not fit beside 128 KB of cells).  Every sketch against the oracle, also with foreign
    bytes and as several records per sketch."""
    rng = np.random.default_rng(9800 + seed)
    K, S, W, H = (31, 15, 12, 4) if seed % 3 == 0 else (int(rng.integers(17, 32)), 15, int(rng.integers(11, 13)), int(rng.integers(2, 6)))
    p = po.make_params(K, S, W, H, 0.0)
    e = native.Engine(K=K, S=S, W=W, H=H)
    recs = [random_record(rng, int(rng.integers(500, 12000)), dirty=(i % 5 == 0)) for i in range(24)]
    recs += [random_record(rng, int(rng.integers(0, 40)), False), random_record(rng, 450, False)]
    sk = e.sketch(recs)
    # ... and virus- / plasmid-sized records (the 1024-thread shapes in front of the same launch), a batch of their own
    big = [random_record(rng, L, dirty=(L == 60_000)) for L in (20_000, 60_000, 150_000, 300_000)]
    skb = e.sketch(big)
    for i, r in enumerate(big):
        assert np.array_equal(skb[i], po.compute_sketch(p, r)), (K, S, W, H, r.size)
    off = np.zeros(len(recs) + 1, np.uint64)
    off[1:] = np.cumsum([r.size for r in recs])
    exp = po.sketch_batch(p, np.concatenate(recs), off)
    bad = [i for i in range(len(recs)) if not np.array_equal(sk[i], exp[i])]
    assert not bad, (K, S, W, H, bad[:8], [recs[i].size for i in bad[:8]])
    er = np.arange(0, len(recs) + 1, 2, dtype=np.uint32)
    sk2 = e.sketch(recs, entry_rec=er)
    for j in range(len(er) - 1):
        acc = np.full(1 << S, -1, np.int32)
        for r in recs[er[j]:er[j + 1]]:
            po.sketch_accumulate(p, r, acc)
        assert np.array_equal(sk2[j], po.densify(p, acc)[0]), (K, S, W, H, j)
    e.close()


@pytest.mark.parametrize("S,W", [(12, 10), (15, 12), (14, 12)])
def test_launch_shape_boundaries(native, po, S, W):
    """Batches of ONE record length on either side of every length at which launch_sketch changes the kernel shape, the
    entry list or the densification (200 / 201: the entry list of the one-wavefront kernel; 415 / 416: one-wavefront or
    workgroup kernel; 16 383 / 16 384: 256 or 1024 threads; 2^18, 2^19 - 1 / 2^19: the distinct-value launch of its own;
    2^21: the chunk length) -- every sketch against the oracle."""
    rng = np.random.default_rng(77 + S)
    p = po.make_params(31, S, W, 4, 0.0)
    e = native.Engine(K=31, S=S, W=W, H=4)
    for L in (199, 200, 201, 222, 223, 414, 415, 416, 450, 16383, 16384, 16385, (1 << 18) - 1, 1 << 18, (1 << 19) - 1, 1 << 19,
              (1 << 21) - 1, (1 << 21) + 1):
        recs = [random_record(rng, L, dirty=(j == 2)) for j in range(3 if L < 100000 else 2)]
        sk = e.sketch(recs)
        for r, got in zip(recs, sk):
            assert np.array_equal(got, po.compute_sketch(p, r)), (S, W, L)
    e.close()
